#!/bin/bash
# Same-box A/B of one environment switch on the captured headline step: tools/ab_env.sh VAR=value [H W classes batch]
# (alternates default / switched, three times; prints ms per step)
SW=$1; shift
for rep in 1 2 3; do
  python tools/ab_step.py "$@"
  env "$SW" python tools/ab_step.py "$@"
done
