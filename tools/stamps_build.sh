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
case $W in rows) F=conv_i8_rows; D=ROWS_STAMPS;; patch) F=conv_i8_patch; D=PATCH_STAMPS;; abl) F=conv_i8_rows; D=ROWS_ABL=$2; W=abl_$2;; split) F=conv_f32_split; D=SPLIT_ABL=$2; W=abl_split_$2;; splitstamps) F=conv_f32_split; D=SPLIT_STAMPS; W=split;; i8m) F=conv_i8; D=I8M_ABL=$2; W=abl_i8m_$2;; *) echo "rows | patch | abl N | split N | splitstamps | i8m N"; exit 1;; esac
[ -f lib/libnna_mars.so ] || bash build.sh
mkdir -p lib/diag
HIPFLAGS="--offload-arch=gfx950 -O3 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -fPIC -std=c++17 -Wno-unused-result -I../include -Icsrc -Icsrc/host"
/opt/rocm/bin/hipcc $HIPFLAGS -D$D -c csrc/hip/$F.hip -o /tmp/${F}_stamps.o
objs=$(ls build/*.o | grep -v "/$F.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o ../thingino-accel_amd/lib/diag/lib_$( [ ${W#abl} != $W ] && echo $W || echo stamps_$W ).so $objs /tmp/${F}_stamps.o -lm
echo built thingino-accel_amd/lib/diag/lib_$( [ ${W#abl} != $W ] && echo $W || echo stamps_$W ).so
