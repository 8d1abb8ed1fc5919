#!/bin/bash
# Like tools/abn.sh, for any stage: bash tools/abn_stage.sh "<v1> <v2> ..." "<stage1> <stage2> ..." [N]
VS=$1; STAGES=$2; N=${3:-4}
for i in $(seq $N); do
  for v in $VS; do
    L=$v; [ "$v" = "default" ] && L=""
    AMDAT_LIB=$L timeout 120 python tools/pipeline_once.py 256 3 64 2>&1 | grep "stages" | python -c "
import sys, ast
l = sys.stdin.read()
d = ast.literal_eval(l[l.index('{'):])
print('$v', ' '.join('%s=%.3f' % (k, d[k]) for k in '$STAGES'.split()), 'sum=%.3f' % sum(d.values()))"
  done
done | sort
