#!/bin/bash
# Builds atomai_amd/lib/libatomai_amd_alt.so: the product objects with ONE source recompiled with extra -D flags
# (for in-process A/B of compile-time variants: tools/gpu_lib_ab.py).   usage: build_alt_lib.sh wgrad "-DAMX_WGRAD_WAVES=1"
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" > /dev/null
SRC=$1; shift
OBJ=atomai_amd/lib/obj
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I atomai_amd/csrc -I include "$@" -c atomai_amd/csrc/$SRC.hip -o /tmp/alt_$SRC.o
OBJS=$(ls $OBJ/*.o | grep -v "/$SRC.o")
hipcc --offload-arch=gfx950 -shared -o atomai_amd/lib/libatomai_amd_alt.so $OBJS /tmp/alt_$SRC.o
echo built atomai_amd/lib/libatomai_amd_alt.so
