B="python bench.py --no-cpu-baseline --no-roofline --no-clean --steps 5 --warmup 1"
P='import sys,json; r=json.loads(sys.stdin.read()); print(r["value"], r["stage_ms_per_step"]["fit_quads"])'
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
echo base; $B 2>&1 | tail -1 | python -c "$P"
for f in exp_tmp/lib_*.so; do cp $f isaac_ros_apriltag_amd/libapriltag_amd.so; echo $f; $B 2>&1 | tail -1 | python -c "$P"; done
