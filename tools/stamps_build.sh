#!/bin/bash
# Diagnostic builds with in-kernel s_memtime stamps (never the shipped library):
#   tools/stamps_build.sh rows   -> thingino-accel_amd/lib/diag/lib_stamps_rows.so  (-DROWS_STAMPS in conv_i8_rows.hip; read with tools/rows_stamps.py)
#   tools/stamps_build.sh patch  -> thingino-accel_amd/lib/diag/lib_stamps_patch.so (-DPATCH_STAMPS in conv_i8_patch.hip)
set -e
cd "$(dirname "$0")/../thingino-accel_amd"
W=${1:-rows}
#   tools/stamps_build.sh abl N  -> thingino-accel_amd/lib/diag/lib_abl_N.so (-DROWS_ABL=N: timing-only ablations of conv_i8_rows)
#   tools/stamps_build.sh splitstamps -> thingino-accel_amd/lib/diag/lib_stamps_split.so (-DSPLIT_STAMPS in conv_f32_split.hip; read with tools/split_stamps.py)
#   tools/stamps_build.sh i8m N -> thingino-accel_amd/lib/diag/lib_abl_i8m_N.so (-DI8M_ABL=N: timing-only ablations of conv_i8_mfma)
#   tools/stamps_build.sh split N -> thingino-accel_amd/lib/diag/lib_abl_split_N.so (-DSPLIT_ABL=N: the same for conv_f32_split)
#   tools/stamps_build.sh fpatch  -> thingino-accel_amd/lib/diag/lib_stamps_fpatch.so (-DFPATCH_STAMPS in conv_f32_patch.hip; read with tools/fpatch_stamps.py)
#   tools/stamps_build.sh fpabl N -> thingino-accel_amd/lib/diag/lib_abl_fpatch_N.so (-DFPATCH_ABL=N: timing-only ablations of conv_f32_patch)
#   tools/stamps_build.sh fpabls N -> thingino-accel_amd/lib/diag/lib_abl_fpatch_sN.so (the same WITH the stamps: LIB=... python tools/fpatch_stamps.py)
#   tools/stamps_build.sh rgbabl 1 | patchabl 1 -> lib_abl_rgb_1.so (conv_i8_rgb without its stores) | lib_abl_patch_1.so (conv_i8_patch without its patch fetch)
case $W in rows) F=conv_i8_rows; D=ROWS_STAMPS;; patch) F=conv_i8_patch; D=PATCH_STAMPS;; abl) F=conv_i8_rows; D=ROWS_ABL=$2; W=abl_$2;; split) F=conv_f32_split; D=SPLIT_ABL=$2; W=abl_split_$2;; splitstamps) F=conv_f32_split; D=SPLIT_STAMPS; W=split;; i8m) F=conv_i8; D=I8M_ABL=$2; W=abl_i8m_$2;; fpatch) F=conv_f32_patch; D=FPATCH_STAMPS; W=fpatch;; fpabl) F=conv_f32_patch; D=FPATCH_ABL=$2; W=abl_fpatch_$2;; fpdef) F=conv_f32_patch; D="$2"; W=abl_fpatch_$3;; rgbabl) F=conv_i8_stem; D=RGB_ABL=$2; W=abl_rgb_$2;; patchabl) F=conv_i8_patch; D=PATCH_ABL=$2; W=abl_patch_$2;; fpabls) F=conv_f32_patch; D="FPATCH_ABL=$2 -DFPATCH_STAMPS"; W=abl_fpatch_s$2;; *) echo "rows | patch | abl N | split N | splitstamps | i8m N | fpatch | fpabl N"; exit 1;; esac
[ -f lib/libnna_mars.so ] || bash build.sh
mkdir -p lib/diag
HIPFLAGS="--offload-arch=gfx950 -O3 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -fPIC -std=c++17 -Wno-unused-result -I../include -Icsrc -Icsrc/host"
/opt/rocm/bin/hipcc $HIPFLAGS -D$D -c csrc/hip/$F.hip -o /tmp/${F}_stamps.o
objs=""  # the objects of the CURRENT sources (as build.sh links them), minus the one rebuilt with the define
for f in csrc/hip/*.hip; do b=$(basename "$f" .hip); [ "$b" = "$F" ] || objs="$objs build/$b.hip.o"; done
for f in csrc/host/*.c; do objs="$objs build/$(basename "$f" .c).o"; done
for f in csrc/host/*.cpp; do objs="$objs build/$(basename "$f" .cpp).o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o ../thingino-accel_amd/lib/diag/lib_$( [ ${W#abl} != $W ] && echo $W || echo stamps_$W ).so $objs /tmp/${F}_stamps.o -lm
echo built thingino-accel_amd/lib/diag/lib_$( [ ${W#abl} != $W ] && echo $W || echo stamps_$W ).so
