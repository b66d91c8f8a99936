#!/bin/bash
# build_head.sh [REV]: complete libeqlb_amd.so of a committed revision (default HEAD) into
# build_exp/lib_head.so, for same-call A/B runs (tools/run_variants.sh).  ALL objects come from that
# revision: mixing objects of two revisions breaks as soon as a struct shared by host and device
# code changes (a mismatched TileDesc made the kernel read out of bounds).
set -e
cd "$(dirname "$0")/.."
REV=${1:-HEAD}
T=$(mktemp -d /tmp/eqlb_head.XXXXXX)
mkdir -p "$T/dolfinx_eqlb_amd/csrc" "$T/include" build_exp
for f in $(git ls-tree --name-only "$REV" dolfinx_eqlb_amd/csrc/ include/); do
  git show "$REV:$f" > "$T/$f"
done
OBJS=""
for f in eqlb_api eqlb_patch_builder eqlb_se_kernels eqlb_projection eqlb_korn eqlb_se_weaksym eqlb_ev eqlb_estimate; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -c "$T/dolfinx_eqlb_amd/csrc/$f.hip" -o "$T/$f.o" &
  OBJS="$OBJS $T/$f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_exp/lib_head.so $OBJS
rm -rf "$T"
echo built build_exp/lib_head.so from $REV
