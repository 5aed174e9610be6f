#!/bin/bash
# Experiment libraries: build/variants/lib_<name>.so from the in-tree sources with extra / replaced hipcc flags.
#   bash scripts/build_variant.sh <name> "<extra -D / -mllvm flags>" [novgpr]
# Only gemm.hip is recompiled per variant (the other objects are cached in build/obj); select at run time with
# UMB_LIB_PATH=build/variants/lib_<name>.so (experiments only: the product always loads umbrella_amd/csrc/libumbrella_hip.so).
root=$(cd "$(dirname "$0")/.." && pwd)
name=$1; extra=$2; mode=$3
src=$root/umbrella_amd/csrc; obj=$root/build/obj; mkdir -p "$obj" "$root/build/variants"
base="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-kernarg-preload-count=16"
vg="-mllvm -amdgpu-mfma-vgpr-form=1"
for f in lowlat gemv epilogue attn sample tp model; do
  if [ ! -f "$obj/$f.o" ] || [ "$src/$f.hip" -nt "$obj/$f.o" ] || [ "$src/common.h" -nt "$obj/$f.o" ]; then
    /opt/rocm/bin/hipcc $base $vg -c "$src/$f.hip" -o "$obj/$f.o" &
  fi
done
g=$vg; [ "$mode" = novgpr ] && g=""
/opt/rocm/bin/hipcc $base $g $extra -c "$src/gemm.hip" -o "$obj/gemm_$name.o" &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "$obj"/{lowlat,gemv,epilogue,attn,sample,tp,model}.o "$obj/gemm_$name.o" -o "$root/build/variants/lib_$name.so" && echo "built build/variants/lib_$name.so"
