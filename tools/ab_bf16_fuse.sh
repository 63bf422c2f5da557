#!/bin/bash
export TMPDIR=/tmp
ROOT=$(pwd)
for f in 0 1; do
rm -rf /tmp/p_ab$f; mkdir -p /tmp/p_ab$f
(cd /tmp && C3D_FUSE_BN_REDUCE_BF16=$f rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_ab$f -o s -- python3 $ROOT/bench.py --matrix-dtype bf16 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events --no-second-engine --no-configs --prewarm 0 --graph off > /dev/null 2>&1)
python3 tools/steady_stats.py $(find /tmp/p_ab$f -name '*kernel_trace.csv' | head -1) gpurun_out/r6_bf16_fuse$f.csv 2
done
