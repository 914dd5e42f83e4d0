#!/bin/bash
# usage: tools/ab.sh "<command>" variant...   -- run the command once per scratch/lib_<variant>.so (same box, same call)
cmd=$1; shift
cp brickmap_amd/libbrickmap_hip.so /tmp/lib_orig.so
for v in "$@"; do
  cp scratch/lib_$v.so brickmap_amd/libbrickmap_hip.so
  echo "== $v"
  eval "$cmd"
done
cp /tmp/lib_orig.so brickmap_amd/libbrickmap_hip.so
