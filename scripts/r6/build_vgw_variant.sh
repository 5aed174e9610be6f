#!/bin/bash
# Experiment libraries for the wide verify GEMM: build/variants/lib_<name>.so = the in-tree objects (umbrella_amd/csrc/build,
# built by __graft_entry__.build()) with vgemm.hip recompiled under extra -D flags.  Select with UMB_LIB_PATH (experiments only).
#   bash scripts/r6/build_vgw_variant.sh <name> "<-D flags>"
root=$(cd "$(dirname "$0")/../.." && pwd)
name=$1; extra=$2
src=$root/umbrella_amd/csrc; mkdir -p "$root/build/variants" "$root/build/obj"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 -mllvm -amdgpu-kernarg-preload-count=16 \
  $extra -c "$src/vgemm.hip" -o "$root/build/obj/vgemm_$name.o" || exit 1
objs=$(ls "$src"/build/*.o | grep -v vgemm.hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs "$root/build/obj/vgemm_$name.o" -o "$root/build/variants/lib_$name.so" && echo "built build/variants/lib_$name.so"
