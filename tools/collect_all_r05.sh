#!/bin/bash
# The round's evidence in three GPU jobs (each under gpurun's 20-minute limit):
#   collect_all_r05.sh 1 : cfg3 and cfg5 through tools/collect_profiles.sh
#   collect_all_r05.sh 2 : cfg3 with diverse group sizes, cfg2 (value records), cfg2 with MSWEEP_DENSE_COMPRESS=0 (the truly
#                          dense sweeps k_dense_passA/B: the review's weak 8)
#   collect_all_r05.sh 3 : set-up stage timings, small inputs, e2e, kernel resources of the sweeps
# Summaries go to gpurun_out/r05_*; copy what is to be judged into profiles/.
set -o pipefail
case "${1:-1}" in
1)
  bash tools/collect_profiles.sh r05_cfg3 > gpurun_out/r05_cfg3.log 2>&1; echo "cfg3 rc=$?"
  bash tools/collect_profiles.sh r05_cfg5 --config cfg5 > gpurun_out/r05_cfg5.log 2>&1; echo "cfg5 rc=$?"
  ;;
2)
  bash tools/collect_profiles.sh r05_diverse --config cfg3 --group-sizes diverse > gpurun_out/r05_diverse.log 2>&1; echo "diverse rc=$?"
  bash tools/collect_profiles.sh r05_cfg2 --config cfg2 > gpurun_out/r05_cfg2.log 2>&1; echo "cfg2 rc=$?"
  MSWEEP_DENSE_COMPRESS=0 bash tools/collect_profiles.sh r05_cfg2_dense --config cfg2 > gpurun_out/r05_cfg2_dense.log 2>&1; echo "cfg2 dense rc=$?"
  ;;
*)
  MSWEEP_BUILD_TIMING=1 python3 bench.py --config cfg3 --steps 5 --warmup 2 --no-cpu-baseline --no-text > gpurun_out/r05_setup_cfg3.json 2> gpurun_out/r05_setup_cfg3.txt; echo "setup cfg3 rc=$?"
  MSWEEP_BUILD_TIMING=1 python3 bench.py --config cfg5 --steps 5 --warmup 2 --no-cpu-baseline --no-text > gpurun_out/r05_setup_cfg5.json 2> gpurun_out/r05_setup_cfg5.txt; echo "setup cfg5 rc=$?"
  python3 tools/small_input_timing.py 100000 1000000 > gpurun_out/r05_small_inputs.txt 2>&1; echo "small rc=$?"
  python3 bench.py --config e2e --no-cpu-baseline --no-text > gpurun_out/r05_e2e_bench_line.json 2> gpurun_out/r05_e2e.err; echo "e2e rc=$?"
  ;;
esac
