set -e
timeout -k 10 300 python -m pytest tests/test_gpu_graph.py -x -q -k "patch_shapes" > gpurun_out/t4.log 2>&1 || { tail -20 gpurun_out/t4.log; exit 1; }
tail -1 gpurun_out/t4.log
timeout -k 10 300 python tools/fpatch_stamps.py D40 L15 L3 D40s2 > gpurun_out/fp_stamps.txt 2>&1 || true
cat gpurun_out/fp_stamps.txt
: > gpurun_out/fp_abl.txt
for n in $ABLS; do echo "== variant $n"; LIB=thingino-accel_amd/lib/diag/lib_abl_fpatch_$n.so timeout -k 10 200 python tools/layer_time.py D40 L15 L3 D40s2 --f32 --cfg f32_mfma=3 --no-oracle 2>&1 | grep -v "^ *$" ; done >> gpurun_out/fp_abl.txt 2>&1
echo "== shipped" >> gpurun_out/fp_abl.txt
timeout -k 10 300 python tools/layer_time.py D40 D20 L15 L3 L35 D80s2 D40s2 D160s2 --f32 --cfg f32_mfma=3 >> gpurun_out/fp_abl.txt 2>&1
cat gpurun_out/fp_abl.txt
