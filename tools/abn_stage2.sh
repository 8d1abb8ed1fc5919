#!/bin/bash
# like abn.sh, for any stage: bash tools/abn_stage2.sh "<v1> <v2> ..." <stage> [N]
VS=$1; ST=$2; N=${3:-5}
for i in $(seq $N); do
  for v in $VS; do
    L=$v; [ "$v" = "default" ] && L=""
    AMDAT_LIB=$L timeout 120 python tools/pipeline_once.py 256 3 64 2>&1 | grep "stages" | python -c "
import sys, ast
l = sys.stdin.read()
d = ast.literal_eval(l[l.index('{'):])
print('$v', round(d['$ST'], 3), round(sum(d.values()), 3))"
  done
done | sort | awk '{a[$1]=a[$1]" "$2; b[$1]=b[$1]" "$3} END {for (k in a) {print k, "'$ST':", a[k]; print k, "sum:", b[k]}}'
