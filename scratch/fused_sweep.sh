#!/bin/bash
# converged-pass duration of the fused kernel for grid sizes and register budgets (scratch: A/B of launch parameters)
for LIBV in default f3; do
  for FB in 256 512 1024 2048; do
    if [ $LIBV = f3 ]; then export SYMMICP_LIB=$PWD/scratch/libs/libsymmicp_f3.so; else unset SYMMICP_LIB; fi
    SYMMICP_FUSED_BLOCKS=$FB timeout -k 10 200 python bench.py --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$LIBV blocks $FB:', d['value'], d['passes']['converged_ms'])"
  done
done
