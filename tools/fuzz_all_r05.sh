#!/bin/bash
# Randomised parity sweeps of the round (GPU box): fuzz_all_r05.sh <part>; part 1: fuzz_parity seeds 7 99 2024; part 2: fuzz_parity
# seeds 31337 1234 + fuzz_build + fuzz_bootstrap.  Last lines go to gpurun_out/r05_fuzz_summary_<part>.txt.
out=gpurun_out/r05_fuzz_summary_${1:-1}.txt
: > $out
run() { echo "== $*" >> $out; timeout -k 10 560 python3 "$@" > gpurun_out/fuzz_tmp.log 2>&1; rc=$?; grep -E "oracles apart|degenerate|FAILED" gpurun_out/fuzz_tmp.log | tail -20 >> $out; tail -1 gpurun_out/fuzz_tmp.log >> $out; echo "rc $rc" >> $out; }
if [ "${1:-1}" = "1" ]; then
  for s in 7 99 2024; do run tools/fuzz_parity.py 300 $s; done
else
  for s in 31337 1234; do run tools/fuzz_parity.py 300 $s; done
  run tools/fuzz_build.py 300 5
  run tools/fuzz_bootstrap.py 100 5
  # the sweeps with the residual correction of the c / z division (1 ulp instead of 2; tools/ab_build.py divcorrect
  # "-DMSW_DIV_CORRECT=1"): kept alive by one short sweep per round (advisor, round 4)
  if [ -f build_ab/lib_divcorrect.so ]; then
    echo "== MSWEEP_CORE_LIB=build_ab/lib_divcorrect.so (-DMSW_DIV_CORRECT=1)" >> $out
    MSWEEP_CORE_LIB=build_ab/lib_divcorrect.so run tools/fuzz_parity.py 150 11
  fi
fi
cat $out
