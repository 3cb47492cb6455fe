#!/bin/bash
# conv_f32_split, three piece products: which phase holds the K step?  (diagnostic builds: tools/stamps_build.sh split N, N a bit set:
# 1 no MFMAs, 2 no gather loads, 4 no split / LDS writes of the input, 8 no fragment reads, 16 no SiLU, 32 no stores).
# GPU box, through gpurun:   bash tools/experiments/r04_f32_split_ablations.sh 0 16 32 48 63
for n in ${@:-0 1 2 4 8 6 14 15 16 32 48 63}; do
  if [ $n = 0 ]; then L=thingino-accel_amd/lib/libnna_mars.so; else L=thingino-accel_amd/lib/diag/lib_abl_split_$n.so; fi
  echo "== SPLIT_ABL=$n"
  LIB=$L timeout -k 10 120 python3 tools/layer_time.py --f32 --no-oracle ${LAYERS:-D40 L15} --cfg f32_mfma=3 | grep f32_mfma
done
