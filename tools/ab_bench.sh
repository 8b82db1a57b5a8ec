#!/bin/bash
# usage: ab.sh lib1 lib2 ...  -> bench each (no cpu baseline), print ms/step and kernel times
for lib in "$@"; do
  if [ "$lib" = "default" ]; then unset MSWEEP_CORE_LIB; else export MSWEEP_CORE_LIB=$lib; fi
  python bench.py --no-cpu-baseline --no-text 2> gpurun_out/ab_err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernels'].items() if k.endswith('ms')})" || tail -5 gpurun_out/ab_err.log
done
