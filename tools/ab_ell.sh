#!/bin/bash
# Development aid, on the GPU box: bash tools/ab_ell.sh LOG v1 v2 ... — rocprofv3 kernel-trace average of the
# log-likelihood kernel for each variants/lib_<v>.so (tools/build_ell_variant.sh) on the cfg3_snps workload
LOG=$1; shift
export TMPDIR=/tmp
for v in "$@"; do
  rm -rf gpurun_out/_kt
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/_kt -o kt -- python3 tools/bench_variant.py variants/lib_$v.so --workload cfg3_snps --steps 3 --warmup 1 --no-cpu-baseline $EXTRA > /dev/null 2> gpurun_out/_kt.err
  echo "== $v" >> $LOG
  find gpurun_out/_kt -name "*kernel_stats.csv" | head -1 | xargs grep ell_kernel | python3 -c "
import sys,csv
for r in csv.reader(sys.stdin):
    print('   ', r[0][-60:], 'avg ms', round(float(r[3])/1e6,3))" >> $LOG
done
rm -rf gpurun_out/_kt
cat $LOG
