#!/bin/bash
# throughput / latency of the resident graph + tail over the batch size (one MI355X); run through gpurun from the repo root:
#   tools/batch_sweep.sh > gpurun_out/batch_sweep.jsonl     (one bench.py line per batch)
set -e
for b in 1 2 4 8 16 32 64 128 256 320 512; do
  steps=$(( b < 32 ? 200 : 40 ))
  python3 bench.py --timed-only --no-cpu-baseline --batch $b --steps $steps --warmup 10
done
