#!/bin/bash
# Copies the summaries tools/profile_r06.sh left under gpurun_out/prof_r06_<cfg>/ into profiles/ (the tracked evidence):
#   tracks -> r06_bench_b1024_*, config4 -> r06_ba_config4_b256_*, sgbm -> r06_reference_pipeline_b256_*; counter JSONs -> profiles/{traffic*.json,orb_valu.json}
set -eu
cd "$(dirname "$0")/.."
for CFG in ${@:-tracks config4 sgbm}; do
  case $CFG in
    tracks)  NAME=r06_bench_b1024;              TJ=traffic_tracks.json;;
    config4) NAME=r06_ba_config4_b256;          TJ=traffic.json;;
    sgbm)    NAME=r06_reference_pipeline_b256;  TJ=traffic_sgbm.json;;
  esac
  D=gpurun_out/prof_r06_$CFG
  [ -d $D ] || { echo "no $D"; continue; }
  cp $D/trace/bench_kernel_stats.csv profiles/${NAME}_kernel_stats.csv
  cp $D/pmc_summary.json profiles/${NAME}_pmc.json
  cp $D/bench.json profiles/${NAME}_profiled_run.json
  { echo "# tools/profile_r06.sh $CFG: rocprofv3 --kernel-trace --stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE, SQ pass (separate runs) over python bench.py --steps 3 --warmup 1 --repeats 1 --in-flight 1 ...";
    cat $D/summary.txt; echo "== SQ counters =="; cat $D/sq_summary.txt; } > profiles/${NAME}_rocprof_summary.txt
  cp $D/$TJ profiles/$TJ
  [ $CFG = tracks ] && [ -f $D/orb_valu.json ] && cp $D/orb_valu.json profiles/orb_valu.json
  [ $CFG = tracks ] && [ -f $D/traffic_orb.json ] && cp $D/traffic_orb.json profiles/traffic_orb.json
  echo "$CFG -> profiles/${NAME}_*, profiles/$TJ"
done
