#!/bin/bash
mkdir -p gpurun_out; make -C oracle -j8 >/dev/null 2>&1
timeout -s KILL 400 python -m pytest tests/test_gpu_map.py -m gpu -x -q --timeout=300 --timeout-method=thread > gpurun_out/map.log 2>&1; echo "rc=$?" >> gpurun_out/map.log
tail -c 3000 gpurun_out/map.log
