#!/bin/bash
# cfg3 with pass B's 8-row / 16-wavefront instantiation forced on the 16-row layout (MSWEEP_PASSB_RC=8: slices of more than
# 8 rows take its streaming branch), two pairs in one GPU job
run() { python bench.py --config cfg3 --no-cpu-baseline --no-text --bootstrap-per-rank 0 --steps 40 --warmup 10 2> gpurun_out/ab_err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernels'].items() if k.endswith('ms')}, d['layout']['passB_reg_cells'])"; }
run base
MSWEEP_PASSB_RC=8 run rc8
run base
MSWEEP_PASSB_RC=8 run rc8
