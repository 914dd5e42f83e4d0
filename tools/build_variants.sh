#!/bin/bash
# usage: tools/build_variants.sh name1 "EXTRA flags 1" name2 "EXTRA flags 2" ...  -> scratch/lib_<name>.so
# (the default library is rebuilt at the end, also when a variant fails to compile)
cd "$(dirname "$0")/.."
mkdir -p scratch
restore() {
  touch brickmap_amd/csrc/*.hip brickmap_amd/csrc/*.cpp
  make -C brickmap_amd/csrc > /dev/null
}
trap restore EXIT
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  touch brickmap_amd/csrc/*.hip brickmap_amd/csrc/*.cpp
  if make -C brickmap_amd/csrc EXTRA="$flags" > /dev/null 2> scratch/build_$name.err; then
    cp brickmap_amd/libbrickmap_hip.so scratch/lib_$name.so
    echo "built $name: $flags"
  else
    echo "FAILED $name: $flags ($(grep -m1 -i 'error\|unknown' scratch/build_$name.err))"
  fi
done
