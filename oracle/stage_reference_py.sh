#!/bin/bash
# oracle/stage_reference_py.sh -- TEST INFRASTRUCTURE.  Stages the reference's own Python driver code (train.py, extract_mesh.py,
# render.py and the packages they import: arguments/, scene/, utils/, gaussian_renderer/) from where it lies under /root/reference
# into oracle/_ref/refpy/ -- git-ignored like the rest of oracle/_ref (nothing of it enters the repository's history), but part of
# the snapshot that travels to the GPU box, where /root/reference does not exist.  The files are byte-identical copies:
# tests/test_e2e_scripts_gpu.py runs them UNCHANGED through launch/run_reference_script.py (SURVEY.md 7 step 6, rows a5 / a21 of 8).
# A manifest of sha256 sums is written beside them so the test can show that what ran is what the reference ships.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REF=/root/reference
OUT="$HERE/_ref/refpy"
[ -f "$REF/train.py" ] || { echo "stage_reference_py.sh: $REF not present, skipping"; exit 0; }
rm -rf "$OUT"
mkdir -p "$OUT"
for f in train.py extract_mesh.py render.py; do cp "$REF/$f" "$OUT/$f"; done
for d in arguments scene utils gaussian_renderer; do
  mkdir -p "$OUT/$d"
  cp "$REF/$d"/*.py "$OUT/$d/"
done
( cd "$OUT" && find . -name '*.py' | sort | xargs sha256sum > MANIFEST.sha256 )
echo "staged $(find "$OUT" -name '*.py' | wc -l) reference Python files into $OUT"
