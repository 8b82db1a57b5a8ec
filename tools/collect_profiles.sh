#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root: bench line + rocprofv3 kernel stats + separate PMC passes
# (HBM traffic, LDS / VALU counters) of a bench workload.  usage: collect_profiles.sh <tag> [bench.py arguments, e.g.
# --config cfg5].  Outputs land in gpurun_out/<tag>/; copy the summaries into profiles/ afterwards
# (tools/pmc_traffic.py makes the traffic JSON).
set -o pipefail
tag=${1:-prof}
shift
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 bench.py "$@" --steps 20 --warmup 5 > $out/bench_line.json 2> $out/bench.err || exit 1
# kernel stats of the SAME command the driver runs (the CPU baseline leg, which launches no kernel, left out)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py "$@" --steps 20 --warmup 5 --no-cpu-baseline --no-text > $out/stats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/pmc/fetch --output-format csv -- python3 bench.py "$@" --steps 20 --warmup 2 --no-cpu-baseline --no-text --no-extras > $out/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/pmc/write --output-format csv -- python3 bench.py "$@" --steps 20 --warmup 2 --no-cpu-baseline --no-text --no-extras > $out/pmc_write.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum -d $out/pmc/tcc --output-format csv -- python3 bench.py "$@" --steps 20 --warmup 2 --no-cpu-baseline --no-text --no-extras > $out/pmc_tcc.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU -d $out/pmc/sq --output-format csv -- python3 bench.py "$@" --steps 20 --warmup 2 --no-cpu-baseline --no-text --no-extras > $out/pmc_sq.log 2>&1 || echo "sq counters failed (non-fatal)"
python3 tools/pmc_traffic.py $out/pmc $out/traffic.json > /dev/null
cp $(ls $out/stats/*/*kernel_stats.csv | head -1) $out/kernel_stats.csv
tail -c 400 $out/bench_line.json
