#!/bin/bash
# development: sweep tile overrides of the decoder's 2-GEMM ResidualUnits / ConvTranspose layers
mkdir -p gpurun_out
out=gpurun_out/decode_sweep.log; : > $out
run() { echo "== $*" >> $out; env "$@" python scripts/decode_time.py >> $out 2>&1 || exit 1; }
run X=0
for t in 256,128,3,32 128,128,3,32 128,128,2,64 256,128,2,64 256,128,3,64; do run DSN_RU1_TILE=$t; done
for t in 256,128,3,32 256,128,3,64; do run DSN_RU7_TILE=$t; done
for t in 256,128,3,32 128,128,3,32 256,256,2,64; do run DSN_CONVT_TILE=$t; done
for t in 256,128,3,32 256,128,3,64; do run DSN_CONVT_DEEP_TILE=$t; done
