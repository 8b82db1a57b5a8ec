#!/bin/bash
# usage: ab_cfgs.sh "cfgA|cfgB|..." lib1 lib2 ... -> ms/step, sweep times and iterations to convergence per (bench arguments, library)
# in ONE GPU job ("default" = the in-tree build; tools/ab_build.py makes the variants); e.g.
#   ab_cfgs.sh "--config cfg3|--config cfg5" default build_ab/lib_x.so
IFS='|' read -ra CFGS <<< "$1"
shift
for cfg in "${CFGS[@]}"; do
  for lib in "$@"; do
    if [ "$lib" = "default" ]; then unset MSWEEP_CORE_LIB; else export MSWEEP_CORE_LIB=$lib; fi
    python bench.py $cfg --no-cpu-baseline --no-text --bootstrap-per-rank 0 --steps 40 --warmup 10 2> gpurun_out/ab_err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d.get('time_to_convergence') or {}; print('$cfg', '$lib', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernels'].items() if k.endswith('ms')}, 'conv', c.get('iters'), c.get('device_ms'))" || tail -5 gpurun_out/ab_err.log
  done
done
