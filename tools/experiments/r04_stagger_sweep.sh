#!/bin/bash
# EXPERIMENT (needs tools/experiments/r04_dual_stagger.patch applied; measured 46.0-47.0k img/s at every stagger against 55.6k joined: not kept): the two half-batches one stagger apart and never re-joined between runs (tuning "dual_stagger" = the launch of the
# first half behind which the second starts; 0 = today's fork / join per run).  GPU box, through gpurun.
for s in ${@:-0 4 8 12 16 24 32 40}; do
  python3 bench.py --timed-only --no-cpu-baseline --steps 40 --warmup 10 --tune dual_stagger=$s | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('dual_stagger=$s', round(d['value']), 'img/s', round(d['ms_per_step'],3), 'ms')"
done
