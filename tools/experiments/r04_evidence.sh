#!/bin/bash
# Raw outputs behind profiles/r04_experiments.md (GPU box, through gpurun; diagnostic libraries from tools/stamps_build.sh):
#   bash tools/experiments/r04_evidence.sh > gpurun_out/r04_evidence.txt 2>&1
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
echo "### conv_i8_rows against the other forms (tools/layer_time.py; variant 20 = rows, 13 = 8-wave implicit GEMM)"
python3 tools/layer_time.py D40 D20 --cfg default --cfg variant=20 --cfg variant=13
echo "### conv_i8_rows ablations on D40 (tools/abl_cycles.sh: 0 shipped, 4 no epilogue, 8 no MFMA, 128 no epilogue / live accumulators, 136 neither, 3 no DMA in the K stream)"
bash tools/abl_cycles.sh D40 0 4 8 128 136 3
cd $R
echo "### copy probe, every form (csrc/probe/mars_probe.hip)"
MARS_PROBE_VERBOSE=1 python3 -c "
import bench
P = bench.load_probe()
print('best %.1f GB/s: %s' % (P.mars_probe_copy_rate_gbs(1 << 30, 10), P.mars_probe_copy_form().decode()))"
echo "### conv_f32_split ablations, three piece products (tools/experiments/r04_f32_split_ablations.sh)"
LAYERS="L0 L15 D40" bash tools/experiments/r04_f32_split_ablations.sh 0 1 2 4 8 16 32 63
echo "### float32 layers, modes 2 (f32 matrix cores) / 4 (bf16 x 6) / 3 (bf16 x 3)"
python3 tools/layer_time.py --f32 L0 L3 L15 L23 D40 P40 --cfg f32_mfma=2 --cfg f32_mfma=4 --cfg f32_mfma=3
