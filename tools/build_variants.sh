#!/bin/bash
# usage: tools/build_variants.sh name1 "flags 1" name2 "flags 2" ...  -> scratch/lib_<name>.so
# flags: -D... options for the compiler (EXTRA of csrc/Makefile), and +<patch> words naming retired experiments under
# tools/variants/<patch>.patch that are applied to the sources for this variant only (e.g. "+lod_pretest -DBM_LOD_PRETEST=1";
# the kernel sources themselves hold no default-off experiment code: tools/variants/README.md).
# (the default library is rebuilt at the end, also when a variant fails to compile)
cd "$(dirname "$0")/.."
mkdir -p scratch
restore() {
  touch brickmap_amd/csrc/*.hip brickmap_amd/csrc/*.cpp
  make -C brickmap_amd/csrc > /dev/null
}
trap restore EXIT
while [ $# -gt 1 ]; do
  name=$1; spec=$2; shift 2
  flags=""; patches=""
  for w in $spec; do
    case $w in +*) patches="$patches ${w#+}";; *) flags="$flags $w";; esac
  done
  ok=1
  for p in $patches; do git apply tools/variants/$p.patch || { echo "FAILED $name: patch $p does not apply"; ok=0; break; }; applied="$applied $p"; done
  touch brickmap_amd/csrc/*.hip brickmap_amd/csrc/*.cpp
  if [ $ok = 1 ] && make -C brickmap_amd/csrc EXTRA="$flags" > /dev/null 2> scratch/build_$name.err; then
    cp brickmap_amd/libbrickmap_hip.so scratch/lib_$name.so
    echo "built $name: $spec"
  elif [ $ok = 1 ]; then
    echo "FAILED $name: $spec ($(grep -m1 -i 'error\|unknown' scratch/build_$name.err))"
  fi
  for p in $applied; do git apply -R tools/variants/$p.patch; done
  applied=""
done
