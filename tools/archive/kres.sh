#!/bin/bash
# kres.sh <source.hip> [extra hipcc flags...] -- per-kernel register usage of a strict-math kernel TU
# (VGPRs, spills, waves/SIMD) as the compiler reports it, one line per kernel.
src=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -DGS_MATH_FUSED=0 \
  -Xclang -fdenormal-fp-math-f32=preserve-sign,ieee -fno-slp-vectorize -S --cuda-device-only \
  -Rpass-analysis=kernel-resource-usage "$@" "$src" -o "${KRES_OUT:-/tmp/kres.s}" 2>&1 |
  grep -E "Function Name|TotalSGPRs|VGPRs:|VGPRs Spill|Occupancy" |
  sed -E 's/.*remark: +//; s/ \[-Rpass.*//' | paste - - - - - | c++filt |
  sed -E 's/Function Name: //; s/\(anonymous namespace\):://; s/\(GsStepArgs[^)]*\)//'
