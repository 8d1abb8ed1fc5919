cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03j
timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r03j/pytest.log 2>&1; echo "pytest exit $?"
tail -5 gpurun_out/r03j/pytest.log
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/r03j/bench.json 2> gpurun_out/r03j/bench.err; echo "bench exit $?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03j/bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value','ms_per_step','n_gpus')}, d.get('roofline'), d.get('cpu_baseline'))
print(d.get('extra', {}).get('stage_ms'), d.get('extra', {}).get('parity_gate'))
PY
