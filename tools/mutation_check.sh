#!/bin/bash
# Shows that the GPU suite is sensitive to the throughput launch set ("a build with k_fit_small stubbed
# out, or with one wrong row constant in k_cc_local<4>, fails pytest -m gpu").  Build the two wrong libraries first (CPU container):
#   python -c "from isaac_ros_apriltag_amd import build as b; b.build_amd_variant('mut1', ['AMDAT_MUTATE=1']); b.build_amd_variant('mut2', ['AMDAT_MUTATE=2'])"
# then on the GPU box:  bash tools/mutation_check.sh > gpurun_out/mutation_check.txt
# Expected: the product build passes the selected tests, both mutants FAIL them.
SEL="throughput_set or (stage_and_detection_parity and c2) or noise_ragged"
for v in "" mut1 mut2; do
  echo "== library: ${v:-product}"
  AMDAT_LIB=$v timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "$SEL" 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" | cut -c1-220
done
