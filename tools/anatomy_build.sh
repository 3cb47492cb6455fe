#!/bin/bash
# Diagnostic build for tools/launch_anatomy.py (never the shipped library): the four int8 convolution translation units with -DANATOMY
# (per-workgroup s_memrealtime stamps into a side buffer: conv_i8_common.hpp) -> thingino-accel_amd/lib/diag/lib_anatomy.so
set -e
cd "$(dirname "$0")/../thingino-accel_amd"
[ -f lib/libnna_mars.so ] || bash build.sh
mkdir -p lib/diag
HIPFLAGS="--offload-arch=gfx950 -O3 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -fPIC -std=c++17 -Wno-unused-result -I../include -Icsrc -Icsrc/host"
AN="conv_i8 conv_i8_patch conv_i8_stem conv_i8_rows"
pids=()
for F in $AN; do /opt/rocm/bin/hipcc $HIPFLAGS -DANATOMY -c csrc/hip/$F.hip -o /tmp/${F}_anat.o & pids+=($!); done
for p in "${pids[@]}"; do wait $p; done
objs=""
for f in csrc/hip/*.hip; do b=$(basename "$f" .hip); case " $AN " in *" $b "*) objs="$objs /tmp/${b}_anat.o";; *) objs="$objs build/$b.hip.o";; esac; done
for f in csrc/host/*.c; do objs="$objs build/$(basename "$f" .c).o"; done
for f in csrc/host/*.cpp; do objs="$objs build/$(basename "$f" .cpp).o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic -o lib/diag/lib_anatomy.so $objs -lm
echo built thingino-accel_amd/lib/diag/lib_anatomy.so
