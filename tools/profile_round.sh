#!/bin/bash
# Profile one bench.py workload on the GPU box; run from the repo root:
#   bash tools/profile_round.sh TAG [extra bench.py flags]
# Writes gpurun_out/TAG_*: kernel-trace stats, then one rocprofv3 --pmc pass per counter set (never combined
# with tracing, as the pool requires), the per-kernel sums (TAG_pmc_summary.txt) and TAG_counters.json — the
# per-read figures bench.py quotes with their provenance.  Copy what should be judged into profiles/.
set -e
TAG=$1; shift || true
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_kt -o kt -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e "$@" > $OUT/${TAG}_bench_under_rocprof.json 2> $OUT/${TAG}_kt.err
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_BRANCH SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $OUT/${TAG}_pmc$i -o pmc -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-e2e "$@" > $OUT/${TAG}_pmc$i.json 2> $OUT/${TAG}_pmc$i.err
  echo "pmc pass $i ($SET) done"
done
python3 tools/pmc_sum.py $OUT ${TAG} > $OUT/${TAG}_pmc_summary.txt
cat $OUT/${TAG}_pmc_summary.txt
