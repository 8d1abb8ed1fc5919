#!/bin/bash
# N-way comparison of builds of the library on the B=256 pipeline: round-robin runs, sorted quad-fit stage times and
# whole-step sums per build.  Usage (on the GPU box): bash tools/abn.sh "<v1> <v2> ..." [N]   ("default" = the product build)
VS=$1; N=${2:-6}
for i in $(seq $N); do
  for v in $VS; do
    L=$v; [ "$v" = "default" ] && L=""
    AMDAT_LIB=$L timeout 120 python tools/pipeline_once.py 256 3 64 2>&1 | grep "stages" | python -c "
import sys, ast
l = sys.stdin.read()
d = ast.literal_eval(l[l.index('{'):])
print('$v', round(d['fit_quads'], 3), round(sum(d.values()), 3))"
  done
done | sort | awk '{a[$1]=a[$1]" "$2; b[$1]=b[$1]" "$3} END {for (k in a) {print k, "fit:", a[k]; print k, "sum:", b[k]}}'
