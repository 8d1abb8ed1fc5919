#!/bin/bash
# fuzz of the round's final tree (CC over the border pass's loser list, cluster list that grows): every dimension of tools/fuzz_gpu.py
f() { timeout $(( $1 + 60 )) python tools/fuzz_gpu.py --cases 1000000 --budget $1 "${@:2}" 2>&1 | grep -E "FAIL|fuzz:" | cut -c1-600; }
f 150 --seed 701
f 100 --seed 702 --batch 5 --maxdim 300
f 100 --seed 703 --batch 12 --maxdim 900
f 100 --seed 704 --colour --layout
f 80 --seed 705 --tile 8 --params
f 100 --seed 706 --maxdim 2600
f 60 --seed 707 --maxdim 90
AMDAT_LIB=stress f 100 --seed 708 --batch 3 --maxdim 500
f 60 --seed 709 --path auto --batch 20 --maxdim 1100
