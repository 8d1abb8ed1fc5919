#!/bin/bash
# Copies what is to be judged from a GPU-box session (gpurun_out/<tag>, written by tools/gpu_round.sh) into profiles/<tag>_*.
#   bash tools/collect_profiles.sh r04_v5 "note for the latency header"
TAG=$1; NOTE=${2:-}
SRC=gpurun_out/$TAG
[ -d $SRC ] || { echo "no $SRC"; exit 1; }
tail -1 $SRC/bench.json > profiles/${TAG}_bench.json
cp $SRC/kernel_stats.md profiles/${TAG}_kernel_stats.md
cp $SRC/fit_timeline.txt profiles/${TAG}_fit_timeline.txt
[ -f $SRC/one_frame_timeline.txt ] && cp $SRC/one_frame_timeline.txt profiles/${TAG}_one_frame_timeline.txt
cp $SRC/pmc/pmc_summary.md profiles/${TAG}_pmc_summary.md
cp $SRC/fq_prof.txt profiles/${TAG}_phase_cycles.txt
cp $SRC/pmc/pipeline_pmc.json profiles/${TAG%%_*}_pipeline_pmc.json
{
  echo "# Single-frame latency and other configs (tools/latency.py, tools/config_rates.py; gpurun_out/$TAG${NOTE:+; $NOTE})"
  echo; echo '```'; cat $SRC/latency.txt; echo '```'; echo; echo '```'; grep -v amdgpu.ids $SRC/config_rates.txt; echo '```'
} > profiles/${TAG}_configs_latency.md
ls -la profiles/${TAG}_*
