#!/bin/bash
# Collects the judged rocprofv3 evidence of one bench configuration on the GPU box:
#   1. --kernel-trace --stats (per-kernel durations)      -> gpurun_out/prof/<tag>_kernel_stats.csv
#   2. --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate runs -> HBM bytes per launch
# and joins them (tools/hbm_report.py).  usage: tools/profile_round.sh <tag> [bench args...]
set -e
TAG=$1; shift
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof
rm -rf /tmp/p_stats /tmp/p_fetch /tmp/p_write      # a stale CSV from an earlier call must not be picked up below
mkdir -p $OUT /tmp/p_stats /tmp/p_fetch /tmp/p_write
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events --no-second-engine --no-configs --prewarm 0 --graph off $@"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -o s -- python3 $ROOT/bench.py $ARGS > $OUT/${TAG}_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p_fetch -o f -- python3 $ROOT/bench.py $ARGS > $OUT/${TAG}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/p_write -o w -- python3 $ROOT/bench.py $ARGS > $OUT/${TAG}_write.log 2>&1
cd $ROOT
S=$(find /tmp/p_stats -name '*kernel_stats.csv' | head -1)
F=$(find /tmp/p_fetch -name '*counter_collection.csv' | head -1)
W=$(find /tmp/p_write -name '*counter_collection.csv' | head -1)
cp $S $OUT/${TAG}_kernel_stats.csv
# steady state: the last two of the four steps only (the --stats summary above also counts what happens once: weight upload,
# first-step packs, optimiser state allocation)
T=$(find /tmp/p_stats -name '*kernel_trace.csv' | head -1)
python3 tools/steady_stats.py $T $OUT/${TAG}_kernel_stats_steady.csv 2 > $OUT/${TAG}_steady.txt
python3 tools/hbm_report.py $F $W $S 4 $OUT/${TAG}_hbm.md $OUT/${TAG}_hbm.json
tail -1 $OUT/${TAG}_stats.log
