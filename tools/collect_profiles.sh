#!/bin/bash
# Copy one round's measurement set from gpurun_out/ (tools/profile_round.sh TAG [...], tools/final_round.sh TAG) into
# profiles/ under the names profiles/README.md lists:  bash tools/collect_profiles.sh r03 r03_ell r03_cfg5
set -e
for TAG in "$@"; do
  G=gpurun_out
  cp $G/${TAG}_kt/kt_kernel_stats.csv profiles/${TAG}_kernel_stats.csv
  cp $G/${TAG}_bench_under_rocprof.json profiles/${TAG}_bench_under_rocprof.json
  cp $G/${TAG}_pmc1/pmc_counter_collection.csv profiles/${TAG}_pmc_pass1_fetch_size.csv
  cp $G/${TAG}_pmc2/pmc_counter_collection.csv profiles/${TAG}_pmc_pass2_write_size.csv
  cp $G/${TAG}_pmc_summary.txt profiles/${TAG}_pmc_summary.txt
  cp $G/${TAG}_counters.json profiles/${TAG}_counters.json
done
T=$1
for f in gpurun_out/${T}_bench_*.json; do
  case $f in *under_rocprof*) continue;; esac
  [ -s "$f" ] && cp $f profiles/
done
ls profiles | grep "^${T}" | wc -l
