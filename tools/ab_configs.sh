#!/bin/bash
# usage: ab_configs.sh lib1 lib2 ... -> ms/step and sweep times of cfg3, cfg3 --group-sizes diverse and cfg5 per library
# ("default" = the in-tree build), one GPU job (tools/ab_build.py makes the variants)
for cfg in "--config cfg3" "--config cfg3 --group-sizes diverse" "--config cfg5"; do
  for lib in "$@"; do
    if [ "$lib" = "default" ]; then unset MSWEEP_CORE_LIB; else export MSWEEP_CORE_LIB=$lib; fi
    python bench.py $cfg --no-cpu-baseline --no-text --no-extras --steps 40 --warmup 10 2> gpurun_out/ab_err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', '$lib', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernels'].items() if k.endswith('ms')})" || tail -5 gpurun_out/ab_err.log
  done
done
