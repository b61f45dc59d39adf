#!/bin/bash
# `downpore overlap` end to end from a FASTA file at BASELINE config 2 (what a user of the command waits for)
mkdir -p gpurun_out/cli
F=/tmp/c2.fa
SECONDS=0
tools/dp_synth reads 2 50000000 100000 10000 0 0 > $F
echo "generate FASTA ${SECONDS}s"; ls -la $F
for i in 1 2; do
  t0=$(date +%s.%N)
  DPH_PROFILE=1 downpore_amd/bin/downpore overlap -input $F -k 13 > /tmp/c2.paf 2> gpurun_out/cli/stderr_$i.txt
  t1=$(date +%s.%N)
  echo "downpore overlap wall $(echo "$t1 - $t0" | bc) s"
  grep "\[cli\]\|\[setup\]\|rounds=" gpurun_out/cli/stderr_$i.txt
  wc -l /tmp/c2.paf; sha256sum /tmp/c2.paf | cut -c1-16
done
