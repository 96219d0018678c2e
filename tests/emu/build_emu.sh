#!/bin/bash
# Builds the CPU-emulated test library from the SAME kernel sources (TEST INFRASTRUCTURE ONLY).
set -e
cd "$(dirname "$0")"
SRC=../../atomai_amd/csrc
OBJS=""
mkdir -p build
for f in $SRC/*.hip; do
  o=build/$(basename $f .hip).o
  if [ ! -f $o ] || [ $f -nt $o ] || [ hip_emu.h -nt $o ] || [ $SRC/amx_device.h -nt $o ]; then
    g++ -O2 -g -std=c++17 -Wno-psabi -fPIC -DAMX_EMU -I. -I$SRC -x c++ -c $f -o $o &
  fi
  OBJS="$OBJS $o"
done
wait
g++ -shared -o libatomai_amd_emu.so $OBJS
echo built tests/emu/libatomai_amd_emu.so
