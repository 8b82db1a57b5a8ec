#!/bin/bash
# the device reader's upload: host threads (MSWEEP_READER_THREADS) x bytes per pinned sub-buffer (MSWEEP_READER_BLOCK_MB);
# every thread streams its own share of the file through two sub-buffers.  First pass of a process and steady passes.
export MSWEEP_PROBE_DIR=${TMPDIR:-/tmp}/msweep_probe_keep
python tools/reader_probe.py 10000000 5000 0 > /dev/null 2>&1
for cfg in "8 2" "16 2" "4 2" "8 1" "8 4" "16 1" "default default" "8 2"; do
  set -- $cfg
  echo "threads $1, block $2 MB"
  if [ "$1" = default ]; then unset MSWEEP_READER_THREADS MSWEEP_READER_BLOCK_MB; else export MSWEEP_READER_THREADS=$1 MSWEEP_READER_BLOCK_MB=$2; fi
  MSWEEP_BUILD_TIMING=1 python tools/reader_probe.py 10000000 5000 0 2>&1 | grep -E "^rep|text to the device" | awk '{printf "%s | ", $0} END {print ""}'
done
rm -rf $MSWEEP_PROBE_DIR
