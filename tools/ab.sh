#!/bin/bash
# usage: tools/ab.sh "<command>" variant...   -- run the command once per scratch/lib_<variant>.so (same box, same call)
cmd=$1; shift
cp brickmap_amd/libbrickmap_hip.so /tmp/lib_orig.so
for v in "$@"; do
  echo "== $v"
  if [ ! -f scratch/lib_$v.so ]; then echo "(no scratch/lib_$v.so: skipped)"; continue; fi
  cp scratch/lib_$v.so brickmap_amd/libbrickmap_hip.so
  eval "$cmd"
done
cp /tmp/lib_orig.so brickmap_amd/libbrickmap_hip.so
