#!/bin/bash
# Development aid, run on the GPU box from the repo root: bash tools/ab.sh LOG v1 v2 ...   (bench each
# variants/lib_<v>.so on the default workload, kernel time of the align launch and reads redone)
LOG=$1; shift
for v in "$@"; do
  echo "== $v" >> $LOG
  timeout -k 10 180 python tools/bench_variant.py variants/lib_$v.so --steps 6 --warmup 2 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['roofline']['kernel_ms_per_launch'],3), d['config']['reads_redone_exact'], d['config']['reads_ok'])" >> $LOG 2>&1
done
cat $LOG
