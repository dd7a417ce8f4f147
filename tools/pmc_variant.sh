#!/bin/bash
# Development aid, on the GPU box from the repo root: bash tools/pmc_variant.sh TAG LIB.so
# SQ counter passes (rocprofv3 --pmc, never together with tracing) over one bench step of a library variant;
# per-kernel sums are written to gpurun_out/TAG_pmc.txt
TAG=$1; LIB=$2
export TMPDIR=/tmp
OUT=gpurun_out
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVES" \
           "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $OUT/${TAG}_pmc$i -o pmc -- python3 tools/bench_variant.py $LIB --steps 1 --warmup 0 --no-cpu-baseline --no-e2e > $OUT/${TAG}_pmc$i.json 2> $OUT/${TAG}_pmc$i.err
  echo "pass $i done"
done
python3 - $OUT $TAG <<'PY' > $OUT/${TAG}_pmc.txt
import sys, glob, csv, collections
out, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(float)
for f in sorted(glob.glob('%s/%s_pmc*/**/*counter_collection.csv' % (out, tag), recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row.get('Kernel_Name', '')
        if 'align3_kernel' not in k: continue
        name = 'rev' if ', 1>' in k else ('fwd' if ', 2>' in k else 'both')
        acc[(name, row['Counter_Name'])] += float(row['Counter_Value'])
for (k, c), v in sorted(acc.items()):
    print('%-5s %-34s %.6g' % (k, c, v))
PY
cat $OUT/${TAG}_pmc.txt
