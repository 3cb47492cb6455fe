run() { b=$1; shift; python3 bench.py --timed-only --batch $b "$@" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1:], round(d['value']), round(d['ms_per_step'],3))" $b "$@"; }
for fw in 128 256 512 1024 2048; do run 128 --tune few_wgs=$fw; done
for fw in 256 512 1024; do run 256 --tune few_wgs=$fw; done
run 128 --tune wres=0; run 128 --tune persist_slots=512; run 128 --tune small_batch=0
