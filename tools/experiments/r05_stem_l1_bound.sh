#!/bin/bash
# Verdict r4 item 3 (stem -> L1 fusion), bounded before it is built: the stem without its stores (lib_abl_rgb_1) and layer 3 without
# its patch fetch (lib_abl_patch_1) -- their sum is what a fused kernel cannot beat (it still has to write the stem's bytes into LDS).
#   bash tools/stamps_build.sh rgbabl 1 && bash tools/stamps_build.sh patchabl 1 && gpurun -- bash tools/experiments/r05_stem_l1_bound.sh
set -e
cd ${GRAFT_REPO_ROOT:-/root/repo}
echo "== shipped"; timeout -k 10 200 python3 tools/layer_time.py L0 L3 --cfg default
echo "== stem, stores dropped"; LIB=thingino-accel_amd/lib/diag/lib_abl_rgb_1.so timeout -k 10 200 python3 tools/layer_time.py L0 --cfg default --no-oracle
echo "== layer 3, patch never fetched"; LIB=thingino-accel_amd/lib/diag/lib_abl_patch_1.so timeout -k 10 200 python3 tools/layer_time.py L3 --cfg default --no-oracle
echo "== copy probe"
MARS_PROBE_VERBOSE=1 timeout -k 10 200 python3 -c "
import bench
P = bench.load_probe()
print('best %.1f GB/s: %s' % (P.mars_probe_copy_rate_gbs(1 << 30, 10), P.mars_probe_copy_form().decode()))"
