#!/bin/bash
# resources of the kernels in one object of the HIP library: LDS bytes, VGPRs, scratch (reads the gfx950 code object's notes)
# usage: tools/kres.sh dp_consensus [libdir]
L=${2:-downpore_amd/lib}
cd $L && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading $1.o > /dev/null 2>&1
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $1.o.0.hipv4-amdgcn-amd-amdhsa--gfx950 | grep -E "\.name:|group_segment_fixed_size|\.vgpr_count|\.sgpr_count|private_segment_fixed" | paste - - - - - | sed -E 's/DpMultiArgs[^ ]*//; s/ +/ /g' | cut -c1-240
rm -f $1.o.0.*
