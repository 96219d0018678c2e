#!/bin/bash
# Builds the CPU-emulated test library from the SAME kernel sources (TEST INFRASTRUCTURE ONLY).
# Safe to call from several processes at once (the multi-process gloo tests do): one builder at a time (flock), objects
# and the library are only rebuilt when stale, and the library is replaced atomically.
set -e
cd "$(dirname "$0")"
SRC=../../atomai_amd/csrc
mkdir -p build
exec 9> build/.lock
flock 9
OBJS=""
NEED_LINK=0
[ -f libatomai_amd_emu.so ] || NEED_LINK=1
for f in $SRC/*.hip; do
  o=build/$(basename $f .hip).o
  STALE=0
  for h in hip_emu.h $SRC/*.h; do [ $h -nt $o ] && STALE=1; done      # any kernel header (conv_kernel.h, ...)
  if [ ! -f $o ] || [ $f -nt $o ] || [ $STALE = 1 ]; then
    g++ -O2 -g -std=c++17 -Wno-psabi -fPIC -DAMX_EMU -I. -I$SRC -x c++ -c $f -o $o &
    NEED_LINK=1
  fi
  OBJS="$OBJS $o"
done
wait
for o in $OBJS; do [ $o -nt libatomai_amd_emu.so ] && NEED_LINK=1; done
# objects of sources that no longer exist must not linger in the link line: OBJS is rebuilt from the sources above
if [ $NEED_LINK = 1 ]; then
  g++ -shared -o build/libatomai_amd_emu.so.tmp $OBJS
  mv -f build/libatomai_amd_emu.so.tmp libatomai_amd_emu.so
  echo built tests/emu/libatomai_amd_emu.so
else
  echo up to date tests/emu/libatomai_amd_emu.so
fi
