#!/bin/bash
# oracle/build_ref.sh -- TEST INFRASTRUCTURE.  Compiles the REFERENCE CUDA rasterizer for gfx950 with
# hipcc, from the sources where they lie under /root/reference, into oracle/_ref/ (git-ignored; the
# built .so travels to the GPU box with the snapshot).  Nothing from the reference is copied into the
# repository: the only transformation is a whitespace normalisation of the CUDA launch chevrons
# ("<< <" -> "<<<", which nvcc accepts and clang does not) and of one brace initialiser clang rejects,
# applied on the fly to temporary files that are deleted after compilation.
#   libgof_cudaref.so        default fp contraction (what an optimising CUDA/HIP compiler would emit)
#   libgof_cudaref_nofma.so  -ffp-contract=off (the oracle's arithmetic contract)
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REF=/root/reference/submodules/diff-gaussian-rasterization
OUT="$HERE/_ref"
[ -d "$REF" ] || { echo "build_ref.sh: $REF not present, skipping"; exit 0; }
mkdir -p "$OUT"
if [ -f "$OUT/libgof_cudaref.so" ] && [ -f "$OUT/libgof_cudaref_nofma.so" ] && [ -f "$OUT/libgof_knnref_nofma.so" ] && [ "$OUT/libgof_cudaref.so" -nt "$HERE/ref_capi.cpp" ] && [ "$OUT/libgof_cudaref.so" -nt "$HERE/../include/gof_hip.h" ] && [ "$OUT/libgof_cudaref.so" -nt "$HERE/build_ref.sh" ] && [ "$OUT/libgof_knnref.so" -nt "$HERE/ref_knn_capi.cpp" ]; then
  echo "oracle/_ref up to date"; exit 0
fi
TMP="$(mktemp -d "$OUT/tmp.XXXXXX")"
trap 'rm -rf "$TMP"' EXIT
INC=(-I "$HERE/ref_shim" -I "$REF/third_party/glm" -I "$REF/cuda_rasterizer" -I "$REF")
for f in forward backward rasterizer_impl; do
  sed -e 's/<< *</<<</g' -e 's/>> *>/>>>/g' -e 's/float2 projected_xy\[MAX_NUM_PROJECTED\] = { 0.f };/float2 projected_xy[MAX_NUM_PROJECTED] = {};/' \
      "$REF/cuda_rasterizer/$f.cu" > "$TMP/$f.cu"
done
build() {   # $1 = suffix, $2... = extra flags
  local sfx="$1"; shift
  local objs=()
  for f in forward backward rasterizer_impl; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w "$@" -x hip "${INC[@]}" -c "$TMP/$f.cu" -o "$TMP/$f$sfx.o" &
    objs+=("$TMP/$f$sfx.o")
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w "$@" -x hip "${INC[@]}" -I "$HERE" -c "$HERE/ref_capi.cpp" -o "$TMP/capi$sfx.o" &
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libgof_cudaref$sfx.so" "${objs[@]}" "$TMP/capi$sfx.o"
}
build "" 
build "_nofma" -ffp-contract=off
echo "built $OUT/libgof_cudaref.so and $OUT/libgof_cudaref_nofma.so"
# the reference's simple-knn (submodules/simple-knn/simple_knn.cu: Morton sort + 1024-point boxes + exact 3-NN), same treatment:
# compiled where it lies, behind the shim headers, with a one-function C wrapper (ref_knn_capi.cpp)
KNN=/root/reference/submodules/simple-knn
if [ -d "$KNN" ]; then
  sed -e 's/<< *</<<</g' -e 's/>> *>/>>>/g' "$KNN/simple_knn.cu" > "$TMP/simple_knn.cu"
  for sfx in "" "_nofma"; do
    fl=(); [ "$sfx" = "_nofma" ] && fl=(-ffp-contract=off)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w "${fl[@]}" -x hip -include cfloat -I "$HERE/ref_shim" -I "$KNN" -c "$TMP/simple_knn.cu" -o "$TMP/simple_knn$sfx.o"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w -x hip -I "$HERE/ref_shim" -I "$KNN" -c "$HERE/ref_knn_capi.cpp" -o "$TMP/knn_capi$sfx.o"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libgof_knnref$sfx.so" "$TMP/simple_knn$sfx.o" "$TMP/knn_capi$sfx.o"
  done
  echo "built $OUT/libgof_knnref.so and $OUT/libgof_knnref_nofma.so"
fi
