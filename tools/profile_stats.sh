#!/bin/bash
# Per-kernel durations of one bench configuration (rocprofv3 --kernel-trace --stats only; tools/profile_round.sh adds the
# HBM counter passes).  usage: tools/profile_stats.sh <tag> [bench args...]   -> gpurun_out/prof/<tag>_kernel_stats.csv
set -e
TAG=$1; shift
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof
rm -rf /tmp/p_stats
mkdir -p $OUT /tmp/p_stats
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events --no-second-engine --no-configs --prewarm 0 --graph off $@"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -o s -- python3 $ROOT/bench.py $ARGS > $OUT/${TAG}_stats.log 2>&1
cd $ROOT
cp $(find /tmp/p_stats -name '*kernel_stats.csv' | head -1) $OUT/${TAG}_kernel_stats.csv
tail -1 $OUT/${TAG}_stats.log | cut -c1-200
