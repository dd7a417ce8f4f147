#!/bin/bash
# End-of-round measurement set, on the GPU box from the repo root:  bash tools/final_round.sh TAG
# bench.py for every workload (gpurun_out/TAG_bench_*.json) — the profiles come from tools/profile_round.sh.
TAG=$1
OUT=gpurun_out
mkdir -p $OUT
run() { name=$1; shift; echo "== $name"; timeout -k 10 600 python bench.py "$@" > $OUT/${TAG}_bench_$name.json 2> $OUT/${TAG}_bench_$name.err || echo "FAILED $name"; tail -c 300 $OUT/${TAG}_bench_$name.json; echo; }
run default
run cfg3_snps --workload cfg3_snps
run cfg4_consensus --workload cfg4_consensus
run cfg5_long --workload cfg5_long --reads 768 --steps 3 --warmup 1
run api_align_signal --workload api_align_signal --no-cpu-baseline
run api_estimate_snps --workload api_estimate_snps --no-cpu-baseline
run api_estimate_snps_notweak --workload api_estimate_snps --no-tweak --no-cpu-baseline
run k10_align --k 10 --reads 4000 --no-cpu-baseline
run k10_snps --workload cfg3_snps --k 10 --reads 4000 --no-cpu-baseline
