#!/bin/bash
# The round's evidence in two GPU jobs (each under gpurun's 20-minute limit):
#   collect_all_r04.sh 1 : cfg3 and cfg5 through tools/collect_profiles.sh
#   collect_all_r04.sh 2 : cfg3 with diverse group sizes, cfg2; set-up stage timings; chain timeline; small inputs; e2e
# Summaries go to gpurun_out/r04_*; copy what is to be judged into profiles/.
set -o pipefail
if [ "${1:-1}" = "1" ]; then
  bash tools/collect_profiles.sh r04_cfg3 > gpurun_out/r04_cfg3.log 2>&1; echo "cfg3 rc=$?"
  bash tools/collect_profiles.sh r04_cfg5 --config cfg5 > gpurun_out/r04_cfg5.log 2>&1; echo "cfg5 rc=$?"
else
  bash tools/collect_profiles.sh r04_diverse --config cfg3 --group-sizes diverse > gpurun_out/r04_diverse.log 2>&1; echo "diverse rc=$?"
  bash tools/collect_profiles.sh r04_cfg2 --config cfg2 > gpurun_out/r04_cfg2.log 2>&1; echo "cfg2 rc=$?"
  MSWEEP_BUILD_TIMING=1 python3 bench.py --config cfg3 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r04_setup_cfg3.json 2> gpurun_out/r04_setup_cfg3.txt; echo "setup cfg3 rc=$?"
  MSWEEP_BUILD_TIMING=1 python3 bench.py --config cfg5 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r04_setup_cfg5.json 2> gpurun_out/r04_setup_cfg5.txt; echo "setup cfg5 rc=$?"
  MSWEEP_CORE_LIB=build_ab/lib_stamps.so python3 tools/chain_timeline.py > gpurun_out/r04_chain_timeline.txt 2>&1; echo "timeline rc=$?"
  python3 tools/small_input_timing.py 100000 1000000 > gpurun_out/r04_small_inputs.txt 2>&1; echo "small rc=$?"
  python3 bench.py --config e2e --no-cpu-baseline > gpurun_out/r04_e2e_bench_line.json 2> gpurun_out/r04_e2e.err; echo "e2e rc=$?"
fi
