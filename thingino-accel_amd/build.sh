#!/bin/bash
# Builds libnna_mars.so (host C + hand-written gfx950 HIP kernels) in-tree.
#   host/*.c   gcc   -O2 -ffp-contract=off   (x86 float semantics of the reference's C)
#   hip/*.hip  hipcc --offload-arch=gfx950 -ffp-contract=off (no FMA contraction in epilogues)
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC="$HERE/csrc"
OUT="$HERE/lib"
OBJ="$HERE/build"
INC="-I$HERE/../include -I$SRC -I$SRC/host"
mkdir -p "$OUT" "$OBJ"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
CC=${CC:-gcc}
CXX=${CXX:-g++}
CFLAGS="-O2 -ffp-contract=off -fPIC -Wall -Wextra -Wno-unused-parameter $INC"
# -amdgpu-mfma-vgpr-form: MFMA results land in VGPRs (gfx950 has one unified file), so the epilogue needs no
# v_accvgpr_read per value and the kernels allocate fewer registers (measured: +1 wave/SIMD on most tiles)
HIPFLAGS="--offload-arch=gfx950 -O3 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -fPIC -std=c++17 -Wno-unused-result $INC"
pids=()
for f in "$SRC"/hip/*.hip; do
  o="$OBJ/$(basename "$f" .hip).hip.o"
  stale=0
  for h in "$SRC"/mhip.h "$SRC"/expf_exact.h "$SRC"/hip/*.hpp; do  # every header a kernel file may include
    [ -f "$h" ] && [ "$h" -nt "$o" ] && stale=1
  done
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ $stale = 1 ]; then
    $HIPCC $HIPFLAGS -c "$f" -o "$o" & pids+=($!)
  fi
done
for f in "$SRC"/host/*.c; do
  o="$OBJ/$(basename "$f" .c).o"
  $CC $CFLAGS -c "$f" -o "$o"
done
for f in "$SRC"/host/*.cpp; do
  o="$OBJ/$(basename "$f" .cpp).o"
  # -ffp-contract=off like the C files: the compile step's BatchNorm fold / quantisation must not turn into FMAs
  $CXX -O2 -ffp-contract=off -fPIC -std=c++17 -Wall -Wextra -Wno-unused-parameter $INC -c "$f" -o "$o"
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
# link exactly the objects of the current sources (a renamed or removed source must not leave its old object in the library)
objs=()
for f in "$SRC"/hip/*.hip; do objs+=("$OBJ/$(basename "$f" .hip).hip.o"); done
for f in "$SRC"/host/*.c; do objs+=("$OBJ/$(basename "$f" .c).o"); done
for f in "$SRC"/host/*.cpp; do objs+=("$OBJ/$(basename "$f" .cpp).o"); done
$HIPCC --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o "$OUT/libnna_mars.so" "${objs[@]}" -lm
echo "built $OUT/libnna_mars.so"
# bench.py's measurement probes (copy rate, shader clock): their own object, not part of the product library
if [ ! -f "$OUT/libmars_probe.so" ] || [ "$SRC/probe/mars_probe.hip" -nt "$OUT/libmars_probe.so" ]; then
  $HIPCC --offload-arch=gfx950 -O3 -fPIC -std=c++17 -shared -o "$OUT/libmars_probe.so" "$SRC/probe/mars_probe.hip"
fi
echo "built $OUT/libmars_probe.so"
# the ONNX -> .mars compile step as the reference's command-line tool (host-only: no GPU runtime linked)
$CC -O2 -Wall $INC "$SRC/cli/mars_main.c" "$OBJ/mars_compile.o" -lstdc++ -lm -o "$OUT/mars"
echo "built $OUT/mars"
