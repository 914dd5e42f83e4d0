#!/bin/bash
# usage: tools/build_variants.sh name1 "EXTRA flags 1" name2 "EXTRA flags 2" ...  -> scratch/lib_<name>.so (base lib restored at the end)
set -e
cd "$(dirname "$0")/.."
mkdir -p scratch
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  touch brickmap_amd/csrc/*.hip brickmap_amd/csrc/*.cpp
  make -C brickmap_amd/csrc EXTRA="$flags" > /dev/null
  cp brickmap_amd/libbrickmap_hip.so scratch/lib_$name.so
  echo "built $name: $flags"
done
touch brickmap_amd/csrc/*.hip brickmap_amd/csrc/*.cpp
make -C brickmap_amd/csrc > /dev/null
