#!/bin/bash
for d in 0 1 3; do echo "DP_SCAN_DEBUG=$d"; DP_SCAN_DEBUG=$d timeout 300 python tools/scan_micro.py 2>&1 | tail -1; done
