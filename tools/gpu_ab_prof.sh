#!/bin/bash
# A/B with the host-side section profile
run() {
  ( cd $1 && DPH_PROFILE=1 timeout 600 python3 bench.py --steps 1 --warmup 1 --cpu-rounds 0 --scan-leg-rounds 0 --dense-leg-rounds 0 --slots ${SLOTS:-8} 2>&1 >/dev/null | grep -E "per executed round" | tail -1 | sed "s/^/$2 /" )
}
for i in 1 2; do run _ab/prev prev; run . new; done
