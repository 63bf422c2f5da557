#!/bin/bash
# The judged rocprofv3 evidence of a round, one capture per (workload, engine) that bench.py reports a roofline for:
#   tools/profile_all.sh   (on the GPU box; ~3 min of box time on a warm image, more on a fresh one)   -> gpurun_out/prof/round6_<workload>_<engine>_{kernel_stats.csv,hbm.md,hbm.json}
# copy the files into profiles/ afterwards.
set -e
cd "$(dirname "$0")/.."
tools/profile_round.sh round6_kitti_bf16x3
tools/profile_round.sh round6_kitti_f32 --matrix-dtype f32
tools/profile_round.sh round6_kitti_bf16 --matrix-dtype bf16
tools/profile_round.sh round6_nuscenes_bf16x3 --height 32 --width 1024 --classes 17 --batch 16
tools/profile_round.sh round6_poss_bf16x3 --height 40 --width 1800 --classes 14 --batch 8 --dataset SemanticPOSS
# matrix-pipe busy fractions of the headline step (one --pmc run, kernel-trace only)
export TMPDIR=/tmp
ROOT=$(pwd)
rm -rf /tmp/p_busy; mkdir -p /tmp/p_busy
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d /tmp/p_busy -o b -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --no-second-engine --no-configs --prewarm 0 --graph off > $ROOT/gpurun_out/prof/round6_busy.log 2>&1)
python3 tools/pmc_step_summary.py $(find /tmp/p_busy -name '*counter_collection.csv' | head -1) 3 > gpurun_out/prof/round6_kitti_bf16x3_mfma_busy.md
