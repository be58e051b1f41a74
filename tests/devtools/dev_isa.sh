#!/bin/bash
# dev_isa.sh <file.hip> [extra flags]: gfx950 assembly of one translation unit with the library's flags -> /tmp/isa/<file>.s,
# plus register / scratch / LDS figures per kernel and an instruction-class histogram.
set -e
HERE=$(cd "$(dirname "$0")/../../gaussian-opacity-fields_amd" && pwd)
f=$1; shift
mkdir -p /tmp/isa
out=/tmp/isa/$(basename "$f" .hip).s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -mllvm -amdgpu-atomic-optimizer-strategy=None \
  -fno-slp-vectorize "$@" --cuda-device-only -S -o "$out" "$HERE/csrc/$f" 2>/dev/null
grep -E "^\s+\.(vgpr_count|sgpr_count|private_segment_fixed_size|group_segment_fixed_size|name):" "$out" | paste - - - - - | sed 's/  */ /g'
echo "--- instruction classes (whole file)"
grep -E "^\s+(v_|s_|ds_|global_|buffer_|flat_)" "$out" | awk '{print $1}' | sed -E 's/_e(32|64)$//' | sort | uniq -c | sort -rn | head -${TOP:-40}
