#!/bin/bash
# kernel cycles (SQ_BUSY_CYCLES / 32 shader engines) and matrix-pipe busy of conv_i8_rows ablation builds on one layer:
#   tools/abl_cycles.sh D40 0 8 128 136     (0 = the shipped library; N = thingino-accel_amd/lib/diag/lib_abl_N.so, tools/stamps_build.sh abl N)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
L=$1; shift
for n in "$@"; do
  O=$R/gpurun_out/ablc_${L}_$n; rm -rf $O; mkdir -p $O
  LIBP=$R/thingino-accel_amd/lib/diag/lib_abl_$n.so; [ "$n" = 0 ] && LIBP=$R/thingino-accel_amd/lib/libnna_mars.so
  LIB=$LIBP rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_WAVES --output-format csv -d $O -o p -- python3 $R/tools/layer_time.py $L --cfg variant=20 --no-oracle > $O.log 2>&1
  python3 - "$O" "$n" <<'PY'
import csv,glob,sys,collections
d=collections.defaultdict(float); n=0
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_i8_rows" in r["Kernel_Name"]:
            d[r["Counter_Name"]]+=float(r["Counter_Value"]); n+= r["Counter_Name"]=="SQ_BUSY_CYCLES"
cyc=d["SQ_BUSY_CYCLES"]/n/32
print("abl %-4s kernel cycles %8.0f  mfma busy %.3f  parked %.2f  VALU/wave %.0f" % (sys.argv[2],cyc,d["SQ_VALU_MFMA_BUSY_CYCLES"]/n/(1024*cyc),d["SQ_WAIT_ANY"]/d["SQ_WAVE_CYCLES"],d["SQ_INSTS_VALU"]/d["SQ_WAVES"]))
PY
done
