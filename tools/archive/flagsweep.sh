#!/bin/bash
# same-box A/B of compiler flags for the step kernels
for flags in "" "-mllvm -amdgpu-sched-strategy=max-ilp" "-mllvm -amdgpu-sched-strategy=max-memory-clause" "-O2" "-mllvm -amdgpu-schedule-relaxed-occupancy=true" "-mllvm -enable-post-misched=false" "-mllvm -amdgpu-enable-max-ilp-scheduling-strategy=1" ""; do
  echo "=== flags: [$flags]"
  if GS_HIP_EXTRA_FLAGS="$flags" python grayscott_amd/_build.py --force > /tmp/build.log 2>&1; then
    python tools/sweep.py --steps 800 --rounds 3 rows_per_block=128,cols_per_lane=2 math=1,rows_per_block=96,cols_per_lane=2 rows_per_block=128,cols_per_lane=4 | tail -3
  else
    tail -3 /tmp/build.log
  fi
done
