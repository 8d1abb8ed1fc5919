#!/bin/bash
# second pass over the final tree (k_cc_border's per-wave append on the throughput set)
f() { timeout $(( $1 + 60 )) python tools/fuzz_gpu.py --cases 1000000 --budget $1 "${@:2}" 2>&1 | grep -E "FAIL|fuzz:" | cut -c1-600; }
timeout 300 python tools/gpu_check.py 2>&1 | grep -E "ALL OK|MISMATCH|FAIL" | head -3
f 120 --seed 801 --path throughput
f 90 --seed 802 --path throughput --batch 5 --maxdim 400
f 90 --seed 803 --path throughput --maxdim 2200
f 60 --seed 804 --path throughput --colour --layout --params
f 60 --seed 805 --path auto --batch 24 --maxdim 1000
AMDAT_LIB=stress f 60 --seed 806 --path throughput --batch 2 --maxdim 600
