# stamps of conv_f32_patch's ablation builds: where does an EMPTY step go?
for n in $ABLS; do echo "== stamps of ablation $n"; LIB=thingino-accel_amd/lib/diag/lib_abl_fpatch_s$n.so timeout -k 10 200 python tools/fpatch_stamps.py D40 L15 L3 2>&1 | grep -v "^ *$"; done > gpurun_out/fp_abl_stamps.txt 2>&1
cat gpurun_out/fp_abl_stamps.txt
