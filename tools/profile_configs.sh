#!/bin/bash
# rocprof evidence for the BASELINE configs OTHER than the headline: config 3 (yolov5n int8 640^2), config 5
# (yolov5s float32 640^2) and the 320^2 workloads north_star names.  Per workload: one `rocprofv3 --kernel-trace --stats` pass and
# separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of the SAME command (one stream, full-batch launches: what bench.py's
# roofline times), summarised by tools/pmc_summary.py (FETCH_SIZE doubled per the guide; tied to the kernel sources by hash), then
# the un-profiled bench line with its CPU leg.   tools/profile_configs.sh TAG [names...]   -> gpurun_out/TAG_<name>_*
set -e
TAG=${1:-r05}; shift || true
NAMES=${@:-"640n f32 320s 320n"}
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
W=2; K=4
for N in $NAMES; do
  case $N in
    640n) ARGS="--width 4"; PW=4; PH=640; FAM=conv_i8; DT=int8;;
    320s) ARGS="--hw 320"; PW=8; PH=320; FAM=conv_i8; DT=int8;;
    320n) ARGS="--hw 320 --width 4"; PW=4; PH=320; FAM=conv_i8; DT=int8;;
    f32)  ARGS="--dtype f32"; PW=8; PH=640; FAM=conv_f32; DT=f32;;
    ship) ARGS="--model tests/golden/models/yolov5n_int8.mars"; PW=0; PH=640; FAM=conv_i8; DT=int8;;  # BASELINE config 3's literal file (NCHW-tagged: the a7 path)
    *) echo "unknown workload $N"; exit 1;;
  esac
  BENCH="bench.py $ARGS --timed-only --steps $K --warmup $W --tune dual_stream_min_batch=0"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_${N}_trace -o t -- python3 $BENCH > $OUT/${TAG}_${N}_line_under_rocprof.json 2> $OUT/${TAG}_${N}_trace.err
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_${N}_pmc_fetch -o f -- python3 $BENCH > /dev/null 2> $OUT/${TAG}_${N}_pmc.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_${N}_pmc_write -o w -- python3 $BENCH > /dev/null 2>> $OUT/${TAG}_${N}_pmc.err
  python3 tools/pmc_summary.py --fetch $OUT/${TAG}_${N}_pmc_fetch --write $OUT/${TAG}_${N}_pmc_write --executions $((W + K)) --width $PW --hw $PH --family $FAM --dtype $DT \
      --note "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of: python3 $BENCH" --out $OUT/${TAG}_${N}_pmc_traffic.json > $OUT/${TAG}_${N}_pmc_traffic.txt
  if [ $N = f32 ]; then  # matrix-pipe occupancy and instruction mix of the float kernels
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA \
        --output-format csv -d $OUT/${TAG}_${N}_pmc_mfma -o m -- python3 $BENCH > /dev/null 2>> $OUT/${TAG}_${N}_pmc.err
    python3 tools/mfma_busy.py $OUT/${TAG}_${N}_pmc_mfma $OUT/${TAG}_${N}_trace $((W + K)) > $OUT/${TAG}_${N}_mfma_busy.json || true
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES \
        --output-format csv -d $OUT/${TAG}_${N}_pmc_mix -o x -- python3 $BENCH > /dev/null 2>> $OUT/${TAG}_${N}_pmc.err
    python3 tools/inst_mix.py $OUT/${TAG}_${N}_pmc_mix $((W + K)) > $OUT/${TAG}_${N}_inst_mix.txt || true
    rm -rf $OUT/${TAG}_${N}_pmc_mfma $OUT/${TAG}_${N}_pmc_mix
  fi
  cp $OUT/${TAG}_${N}_pmc_traffic.json profiles/${TAG}_${N}_pmc_traffic.json  # bench.py reads it (hash-tied) for this workload's roofline.traffic
  find $OUT/${TAG}_${N}_trace -name '*kernel_stats.csv' -exec cp {} $OUT/${TAG}_${N}_kernel_stats.csv \;
  rm -rf $OUT/${TAG}_${N}_trace $OUT/${TAG}_${N}_pmc_fetch $OUT/${TAG}_${N}_pmc_write
  python3 bench.py $ARGS --sustain-s 1 --no-extra-configs $( [ $N = f32 ] && echo "--steps 10 --warmup 3" ) > $OUT/${TAG}_${N}_bench.json 2> $OUT/${TAG}_${N}_bench.err
  echo "$N done: $(python3 -c "import json;d=json.load(open('$OUT/${TAG}_${N}_bench.json'));print(d['value'], d['roofline']['frac'])")"
done
