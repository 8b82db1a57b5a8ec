#!/bin/bash
# the device reader's host threads (MSWEEP_READER_THREADS: pread into the pinned staging): is the upload bound by them?
export MSWEEP_PROBE_DIR=${TMPDIR:-/tmp}/msweep_probe_keep
python tools/reader_probe.py 10000000 5000 0 > /dev/null 2>&1
for t in 4 8 16 32; do
  echo "threads $t"
  MSWEEP_READER_THREADS=$t MSWEEP_BUILD_TIMING=1 python tools/reader_probe.py 10000000 5000 0 2>&1 | grep -E "^rep|text to the device" | tail -6
done
rm -rf $MSWEEP_PROBE_DIR
