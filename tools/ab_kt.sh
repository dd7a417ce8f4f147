#!/bin/bash
# Development aid, on the GPU box: bash tools/ab_kt.sh LOG v1 v2 ... — rocprofv3 kernel-trace averages of the align
# kernels (reverse / forward launches) for each variants/lib_<v>.so on the default workload (EXTRA='--workload cfg5_long
# --reads 768': other bench.py flags)
LOG=$1; shift
export TMPDIR=/tmp
for v in "$@"; do
  rm -rf gpurun_out/_kt
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/_kt -o kt -- python3 tools/bench_variant.py variants/lib_$v.so --steps 3 --warmup 1 --no-cpu-baseline --no-e2e $EXTRA > /dev/null 2> gpurun_out/_kt.err
  echo "== $v" >> $LOG
  find gpurun_out/_kt -name "*kernel_stats.csv" | head -1 | xargs grep align3 | python3 -c "
import sys,csv
for r in csv.reader(sys.stdin):
    print('   ', r[0][-40:], 'avg ms', round(float(r[3])/1e6,3))" >> $LOG
done
rm -rf gpurun_out/_kt
cat $LOG
