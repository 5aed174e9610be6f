#!/bin/bash
# Experiment libraries: build/variants/lib_<name>.so = the in-tree objects (umbrella_amd/csrc/build, written by
# __graft_entry__.build()) with ONE source recompiled under extra / replaced hipcc flags.
#   bash scripts/build_variant.sh <name> "<extra -D / -mllvm flags>" [source.hip (default gemm.hip)] [novgpr]
# Select at run time with UMB_LIB_PATH=build/variants/lib_<name>.so (experiments only: the product always loads
# umbrella_amd/csrc/libumbrella_hip.so).  scripts/r6/build_vgw_variant.sh is the same for vgemm.hip.
root=$(cd "$(dirname "$0")/.." && pwd)
name=$1; extra=$2; srcf=${3:-gemm.hip}; mode=$4
src=$root/umbrella_amd/csrc; mkdir -p "$root/build/variants" "$root/build/obj"
python -c "import sys; sys.path.insert(0, '$root'); import __graft_entry__ as g; g.build()" > /dev/null 2>&1
vg="-mllvm -amdgpu-mfma-vgpr-form=1"; [ "$mode" = novgpr ] && vg=""
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $vg -mllvm -amdgpu-kernarg-preload-count=16 $extra \
  -c "$src/$srcf" -o "$root/build/obj/${srcf%.hip}_$name.o" || exit 1
objs=$(ls "$src"/build/*.o | grep -v "/$srcf\.")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs "$root/build/obj/${srcf%.hip}_$name.o" -o "$root/build/variants/lib_$name.so" \
  && echo "built build/variants/lib_$name.so"
