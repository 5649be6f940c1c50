#!/bin/bash
# Builds oracle/_ref/ref_esti_plane and oracle/_ref/ref_iekf from the reference's own sources (never copied into the repo).
# Exits 0 with an explanation when the reference tree or its dependencies (Eigen3, Boost headers) are not available.
set -u
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${1:-/root/reference}"
OUT="$HERE/../_ref"
if [ ! -f "$REF/include/common_lib.h" ]; then echo "[oracle/ref] no reference tree at $REF: nothing built"; exit 0; fi
EIGEN=""
for d in /usr/include/eigen3 /usr/local/include/eigen3 /opt/homebrew/include/eigen3 "${EIGEN3_INCLUDE_DIR:-/nonexistent}"; do
  [ -f "$d/Eigen/Dense" ] && EIGEN="$d" && break
done
if [ -z "$EIGEN" ]; then
  echo "[oracle/ref] Eigen3 headers not found (looked in /usr/include/eigen3, /usr/local/include/eigen3, \$EIGEN3_INCLUDE_DIR):"
  echo "             the reference cannot be compiled here; oracle/_ref stays empty and tests/test_ref_recipe.py skips."
  exit 0
fi
mkdir -p "$OUT"
# the reference function, taken from where it lies (lines 225-257 = esti_plane<T>), into the build directory only
sed -n '225,257p' "$REF/include/common_lib.h" > "$OUT/esti_plane_ref.inc"
grep -q "colPivHouseholderQr" "$OUT/esti_plane_ref.inc" || { echo "[oracle/ref] common_lib.h:225-257 is not esti_plane any more"; exit 1; }
CXXFLAGS="-std=c++14 -O3"   # the reference's own (CMakeLists.txt:8,14): no -march, no -ffast-math
g++ $CXXFLAGS -I"$EIGEN" -I"$OUT" "$HERE/ref_esti_plane.cpp" -o "$OUT/ref_esti_plane" || exit 1
echo "[oracle/ref] built $OUT/ref_esti_plane (Eigen at $EIGEN)"
BOOST=""
for d in /usr/include /usr/local/include "${BOOST_INCLUDE_DIR:-/nonexistent}"; do
  [ -f "$d/boost/preprocessor/seq.hpp" ] && BOOST="$d" && break
done
if [ -z "$BOOST" ]; then echo "[oracle/ref] Boost headers not found: ref_iekf (esekfom.hpp) not built"; exit 0; fi
g++ $CXXFLAGS -I"$EIGEN" -I"$BOOST" -I"$REF/include" "$HERE/ref_iekf.cpp" -o "$OUT/ref_iekf" || exit 1
echo "[oracle/ref] built $OUT/ref_iekf"
