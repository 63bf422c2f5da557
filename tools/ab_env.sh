#!/bin/bash
# A/B of environment-variable settings on the GPU box: tools/ab_env.sh "VAR=a VAR2=b" "VAR=c" ...  (each arg = one setting)
for setting in "$@"; do
  for rep in 1 2; do
    env $setting python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-second-engine --graph off 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$setting', d['value'], d['ms_per_step'], d['roofline']['all_mfma_kernels']['ms_per_step'])"
  done
done
