#!/bin/bash
# the device reader's staging chunk size (MSWEEP_READER_CHUNK_MB): first pass of a process and steady passes, one GPU job
export MSWEEP_PROBE_DIR=${TMPDIR:-/tmp}/msweep_probe_keep
python tools/reader_probe.py 10000000 5000 0 > /dev/null 2>&1   # generates the strands, warms the page cache
for mb in 16 32 64 128 256 64; do
  echo "chunk ${mb} MB"
  MSWEEP_READER_CHUNK_MB=$mb python tools/reader_probe.py 10000000 5000 0 2>&1 | grep "^rep"
done
rm -rf $MSWEEP_PROBE_DIR
