#!/bin/bash
export MSWEEP_PROBE_DIR=${TMPDIR:-/tmp}/msweep_probe_keep
python tools/reader_probe.py 10000000 5000 0 > /dev/null 2>&1
one() { MSWEEP_BUILD_TIMING=1 python tools/reader_probe.py 10000000 5000 0 2>&1 | grep -E "^rep|text to the device" | awk '{printf "%s | ", $0} END {print ""}' | sed 's/\[msweep reader\/device\] text to the device (first strand) */up /g'; }
for rep in 1 2; do
echo "old gang, 8 threads";  MSWEEP_CORE_LIB=build_ab/lib_oldstager.so MSWEEP_READER_THREADS=8 one
echo "old gang, 16 threads"; MSWEEP_CORE_LIB=build_ab/lib_oldstager.so MSWEEP_READER_THREADS=16 one
echo "streams, 8 x 4 MB"; MSWEEP_READER_THREADS=8 MSWEEP_READER_BLOCK_MB=4 one
echo "streams, 8 x 8 MB"; MSWEEP_READER_THREADS=8 MSWEEP_READER_BLOCK_MB=8 one
echo "streams, 6 x 4 MB"; MSWEEP_READER_THREADS=6 MSWEEP_READER_BLOCK_MB=4 one
done
rm -rf $MSWEEP_PROBE_DIR
