#!/bin/bash
# A/B of two TREES (an old commit against the working tree), each with its own library and bench.py, in one GPU job:
#   git worktree add build_ab/old_tree <commit> && (cd build_ab/old_tree && python __graft_entry__.py)
#   gpurun -- 'bash tools/ab_trees.sh build_ab/old_tree "--config cfg3 --steps 210 --warmup 1" ["more bench args" ...]'
# (an old library cannot be loaded through MSWEEP_CORE_LIB once the C ABI has grown: tools/ab_build.py is for macros).
# --steps = the iterations of the converged solve times the WHOLE trajectory, rejected steps included
# (profiles/r04_vs_r03_same_box.txt).  Remove the worktree afterwards: git worktree remove --force build_ab/old_tree
old=$1
shift
run() { (cd $2 && python bench.py $1 --no-cpu-baseline --no-extras 2> /tmp/ab_err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', '$2', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernels'].items() if k.endswith('ms')})" || tail -5 /tmp/ab_err.log); }
for args in "$@"; do
  for tree in . $old . $old; do run "$args" $tree; done
done
