#!/bin/bash
# Builds atomai_amd/lib/libatomai_amd_<name>.so = the product objects with some sources recompiled under extra -D flags,
# for in-process A/B of compile-time variants (tools/gpu_probe_r02.py, tools/gpu_step_ab.py).
#   usage: build_variant_lib.sh <name> "<flags>" src1 [src2 ...]      e.g.  swp "-DAMX_CONV_SWP=1" conv_fwd_3x3 conv_fwd_dil
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" > /dev/null
NAME=$1; FLAGS=$2; shift 2
OBJ=atomai_amd/lib/obj
EXCL=""
ALT=""
for SRC in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -Wno-pass-failed -I atomai_amd/csrc -I include $FLAGS \
    -c atomai_amd/csrc/$SRC.hip -o /tmp/var_${NAME}_$SRC.o &
  EXCL="$EXCL -e /$SRC.o"
  ALT="$ALT /tmp/var_${NAME}_$SRC.o"
done
wait
OBJS=$(ls $OBJ/*.o | grep -v $EXCL)
hipcc --offload-arch=gfx950 -shared -o atomai_amd/lib/libatomai_amd_$NAME.so $OBJS $ALT
echo built atomai_amd/lib/libatomai_amd_$NAME.so
