#!/bin/bash
# Tuning sweep over the quad-fit size classes ("threads:key-capacity:grid-budget" x 5, see AMDAT_FQ_CLASSES in
# isaac_ros_apriltag_amd/csrc/detector.hip).  Run on the GPU box: bash tools/fq_sweep.sh
for cfg in "64:256:32768,128:1024:16384,256:4096:2048,512:8192:1024,512:16384:512" \
           "64:256:32768,128:1024:16384,256:4096:4096,512:8192:1024,512:16384:512" \
           "64:256:65536,128:1024:32768,256:4096:2048,512:8192:1024,512:16384:512" \
           "64:256:32768,128:1024:16384,256:2048:4096,512:8192:2048,512:16384:512"; do
  echo -n "$cfg -> "
  AMDAT_FQ_CLASSES=$cfg python bench.py --no-cpu-baseline --no-roofline --no-clean --steps 5 2>&1 | tail -1 | \
    python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['stage_ms_per_step']['fit_quads'])"
done
