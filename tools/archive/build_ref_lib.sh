#!/bin/bash
# Builds atomai_amd/lib/libatomai_amd_ref.so from the kernel sources of a git revision (default: the round-1 tag
# commit b022c40) for in-process A/B against the working tree's library (tools/gpu_probe_r02.py).
set -e
cd "$(dirname "$0")/.."
REV=${1:-b022c40}
NAME=${2:-ref}            # atomai_amd/lib/libatomai_amd_<NAME>.so (the revision must have the product library's ABI)
D=/tmp/amx_ref_$REV
rm -rf $D; mkdir -p $D/csrc $D/include $D/obj
for f in $(git ls-tree --name-only $REV atomai_amd/csrc/); do git show $REV:$f > $D/csrc/$(basename $f); done
git show $REV:include/atomai_amd.h > $D/include/atomai_amd.h
for s in $D/csrc/*.hip; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -Wno-pass-failed -I $D/csrc -I $D/include -c $s -o $D/obj/$(basename $s .hip).o &
done
wait
hipcc --offload-arch=gfx950 -shared -o atomai_amd/lib/libatomai_amd_$NAME.so $D/obj/*.o
echo built atomai_amd/lib/libatomai_amd_$NAME.so from $REV
