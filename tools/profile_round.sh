#!/bin/bash
# One profiling pass of bench.py on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh TAG      -> gpurun_out/TAG_*  (copy what should be judged into profiles/)
# rocprofv3 gets the interpreter directly after `--`; PMC passes are separate and carry no trace flags.
set -e
TAG=${1:-r01}
OUT=$PWD/gpurun_out
mkdir -p $OUT
W=2; K=4
# the default launch policy (bench.py's default; --autotune would add ~4 launches of every variant of every layer)
BENCH="bench.py --no-cpu-baseline --steps $K --warmup $W"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -o t -- python3 $BENCH > $OUT/${TAG}_bench_line_under_rocprof.json 2> $OUT/${TAG}_trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -o f -- python3 $BENCH > /dev/null 2> $OUT/${TAG}_pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -o w -- python3 $BENCH > /dev/null 2> $OUT/${TAG}_pmc_write.err
python3 tools/pmc_summary.py --fetch $OUT/${TAG}_pmc_fetch --write $OUT/${TAG}_pmc_write --executions $((W + K + 2)) \
    --note "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of: python3 $BENCH" --out $OUT/${TAG}_pmc_traffic.json > $OUT/${TAG}_pmc_traffic.txt
find $OUT/${TAG}_trace -name '*kernel_stats.csv' -exec cp {} $OUT/${TAG}_bench_kernel_stats.csv \;
rm -rf $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write
find $OUT/${TAG}_trace -name '*kernel_trace.csv' -delete
