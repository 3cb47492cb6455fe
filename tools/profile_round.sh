#!/bin/bash
# One profiling round of bench.py on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh TAG      -> gpurun_out/TAG_*  (copy what should be judged into profiles/)
# rocprofv3 gets the interpreter directly after `--`; PMC passes are separate and carry no trace flags.
set -e
TAG=${1:-r04}
OUT=$PWD/gpurun_out
mkdir -p $OUT
W=2; K=4
# the default launch policy (bench.py's default); only the timed region, no I/O / CPU legs
# The per-kernel passes run every launch over the full batch on ONE stream (--tune dual_stream_min_batch=0), which is
# also how bench.py times the kernels for its roofline object: per-launch averages then mean one launch = one layer of
# 256 frames.  (The default run cuts the batch into two halves on two streams whose launches overlap; its trace is kept
# too, as TAG_two_streams_kernel_stats.csv.)
BENCH="bench.py --timed-only --steps $K --warmup $W --tune dual_stream_min_batch=0"
BENCH2="bench.py --timed-only --steps $K --warmup $W"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -o t -- python3 $BENCH > $OUT/${TAG}_bench_line_under_rocprof.json 2> $OUT/${TAG}_trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -o f -- python3 $BENCH > /dev/null 2> $OUT/${TAG}_pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -o w -- python3 $BENCH > /dev/null 2> $OUT/${TAG}_pmc_write.err
python3 tools/pmc_summary.py --fetch $OUT/${TAG}_pmc_fetch --write $OUT/${TAG}_pmc_write --executions $((W + K)) \
    --note "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of: python3 $BENCH" --out $OUT/${TAG}_pmc_traffic.json > $OUT/${TAG}_pmc_traffic.txt
# matrix-pipe occupancy per kernel symbol: MFMA-busy cycles against the cycles the CUs were busy
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA \
    --output-format csv -d $OUT/${TAG}_pmc_mfma -o m -- python3 $BENCH > /dev/null 2> $OUT/${TAG}_pmc_mfma.err
python3 tools/mfma_busy.py $OUT/${TAG}_pmc_mfma $OUT/${TAG}_trace $((W + K)) > $OUT/${TAG}_mfma_busy.json
# instruction mix per kernel symbol (vector and scalar instructions per MFMA)
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES \
    --output-format csv -d $OUT/${TAG}_pmc_mix -o x -- python3 $BENCH > /dev/null 2> $OUT/${TAG}_pmc_mix.err
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/${TAG}_pmc_lds -o l -- python3 $BENCH > /dev/null 2> $OUT/${TAG}_pmc_lds.err
python3 tools/inst_mix.py $OUT/${TAG}_pmc_mix $((W + K)) --json $OUT/${TAG}_inst_mix.json --lds $OUT/${TAG}_pmc_lds > /dev/null
cp $OUT/${TAG}_inst_mix.json profiles/inst_mix.json  # bench.py reads it (hash-tied like pmc_traffic.json) for roofline.pipes
{ echo "# ${TAG}: instruction mix per convolution kernel symbol"; echo; echo "\`rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES -- python3 $BENCH\` (tools/inst_mix.py; per step = per batch of 256 frames, every launch of the symbol summed)"; echo; python3 tools/inst_mix.py $OUT/${TAG}_pmc_mix $((W + K)); } > $OUT/${TAG}_deep_sq_counters.md
find $OUT/${TAG}_trace -name '*kernel_stats.csv' -exec cp {} $OUT/${TAG}_bench_kernel_stats.csv \;
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace2 -o t -- python3 $BENCH2 > $OUT/${TAG}_two_streams_line_under_rocprof.json 2> $OUT/${TAG}_trace2.err
find $OUT/${TAG}_trace2 -name '*kernel_stats.csv' -exec cp {} $OUT/${TAG}_two_streams_kernel_stats.csv \;
find $OUT/${TAG}_trace2 -name '*kernel_trace.csv' -delete
rm -rf $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_pmc_mfma $OUT/${TAG}_pmc_mix $OUT/${TAG}_pmc_lds
find $OUT/${TAG}_trace -name '*kernel_trace.csv' -delete
# un-profiled lines: per-launch table, config 3 and the 320x320 workloads WITH their CPU-baseline / bit-exact legs (shorter
# sustained leg: these are secondary lines), the float32 workload (config 5), the default line last
python3 bench.py --timed-only --ops $OUT/${TAG}_ops.txt > $OUT/${TAG}_bench_ops.json 2> $OUT/${TAG}_ops.err
python3 bench.py --width 4 --sustain-s 1 --no-extra-configs > $OUT/${TAG}_bench_640_yolov5n.json 2>> $OUT/${TAG}_ops.err
python3 bench.py --hw 320 --sustain-s 1 --no-extra-configs > $OUT/${TAG}_bench_320_yolov5s.json 2>> $OUT/${TAG}_ops.err
python3 bench.py --hw 320 --width 4 --sustain-s 1 --no-extra-configs > $OUT/${TAG}_bench_320_yolov5n.json 2>> $OUT/${TAG}_ops.err
python3 bench.py --dtype f32 --steps 10 --warmup 3 --sustain-s 1 > $OUT/${TAG}_f32_bench.json 2>> $OUT/${TAG}_ops.err
cp $OUT/${TAG}_pmc_traffic.json profiles/pmc_traffic.json  # bench.py reads it (and checks the kernel-source hash inside) for roofline.traffic
python3 bench.py > $OUT/${TAG}_bench_default.json 2>> $OUT/${TAG}_ops.err
