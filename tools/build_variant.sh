#!/bin/bash
# build_variant.sh NAME "-DFOO -DBAR": timing-experiment build of libeqlb_amd.so into build_exp/lib_NAME.so
# (only eqlb_se_kernels.hip is recompiled; run with EQLB_AMD_LIB=build_exp/lib_NAME.so python bench.py)
set -e
cd "$(dirname "$0")/.."
mkdir -p build_exp
SRC=dolfinx_eqlb_amd/csrc
FILE=${3:-eqlb_se_kernels}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $2 -c $SRC/$FILE.hip -o build_exp/${FILE}_$1.o
OBJS=""
for f in eqlb_api eqlb_patch_builder eqlb_se_kernels eqlb_stress_tiled eqlb_projection eqlb_korn eqlb_se_weaksym eqlb_ev eqlb_estimate eqlb_halo_rccl eqlb_tiling_device; do
  if [ "$f" = "$FILE" ]; then OBJS="$OBJS build_exp/${FILE}_$1.o"; else OBJS="$OBJS $SRC/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_exp/lib_$1.so $OBJS -ldl
echo built build_exp/lib_$1.so
