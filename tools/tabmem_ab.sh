set -e
run() { python bench.py --config cfg3 --no-cpu-baseline --no-text --bootstrap-per-rank 0 --steps 40 --warmup 10 2> gpurun_out/ab_err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernels'].items() if k.endswith('ms')}, d['layout']['table_in_lds'], d['layout']['record_bytes'], d['layout']['index_records'])"; }
run base
MSWEEP_FORCE_LDS=10 MSWEEP_HYBRID=0 run tabmem
run base
MSWEEP_FORCE_LDS=10 MSWEEP_HYBRID=0 run tabmem
