#!/bin/bash
# Same-box A/B of two builds of the library on the captured headline step: tools/ab_lib.sh <other libcoarse3d_hip.so> [H W classes batch]
# (alternates default build / other build twice; prints ms per step)
OTHER=$1; shift
for rep in 1 2; do
  for lib in "" "$OTHER"; do
    C3D_LIB=$lib python tools/ab_step.py "$@"
  done
done
