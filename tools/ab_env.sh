#!/bin/bash
# usage: ab_env.sh VAR val1 val2 ...  -> bench.py (no cpu baseline) once per value of the developer switch VAR
# ("-" = unset), printing ms/step and the sweeps' kernel times
var=$1; shift
for v in "$@"; do
  if [ "$v" = "-" ]; then unset $var; else export $var=$v; fi
  python bench.py --no-cpu-baseline --no-text 2> gpurun_out/ab_err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$var=$v', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernels'].items() if k.endswith('ms')})" || tail -5 gpurun_out/ab_err.log
done
