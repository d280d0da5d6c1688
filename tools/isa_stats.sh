#!/bin/bash
# Per-kernel register / scratch / LDS figures of a built translation unit (no GPU needed):
#   tools/isa_stats.sh grayscott_amd/build/gs_step_strict_op.o [name filter]
# and, with DISASM=1, the instruction mix of the kernels that match the filter.
set -eo pipefail
B=/opt/rocm/lib/llvm/bin
OBJ=$1; FILTER=${2:-}
T=$(mktemp -d)
$B/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin "$OBJ"
$B/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/dev.co --unbundle
$B/llvm-readelf --notes $T/dev.co | awk '
  /\.name:/ {name=$2} /\.vgpr_count:/ {v=$2} /\.sgpr_count:/ {s=$2} /\.private_segment_fixed_size:/ {p=$2}
  /\.group_segment_fixed_size:/ {g=$2} /\.vgpr_spill_count:/ {sp=$2; print name, "vgpr", v, "sgpr", s, "scratch", p, "lds", g, "spill", sp}' \
  | while read -r name rest; do echo "$(echo "$name" | c++filt | sed 's/(anonymous namespace):://; s/(GsStepArgs.*//') $rest"; done | grep -E "${FILTER:-.}" || true
if [ -n "$DISASM" ]; then
  $B/llvm-objdump -d $T/dev.co > $T/dev.s
  echo "disassembly: $T/dev.s"
else
  rm -rf $T
fi
