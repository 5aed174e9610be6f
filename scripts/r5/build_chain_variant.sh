#!/bin/bash
# Experiment library: build/variants/lib_chain_<name>.so = the product's objects (umbrella_amd/csrc/build, after
# __graft_entry__.build()) with chain.hip recompiled under extra flags.  bash scripts/r5/build_chain_variant.sh trace "-DUMB_CHAIN_TRACE"
root=$(cd "$(dirname "$0")/../.." && pwd)
name=$1; extra=$2
src=$root/umbrella_amd/csrc; mkdir -p "$root/build/variants"
python -c "import sys; sys.path.insert(0, '$root'); import __graft_entry__ as g; g.build()" > /dev/null
objs=$(ls "$src"/build/*.o | grep -v "chain.hip")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 -mllvm -amdgpu-kernarg-preload-count=16 $extra \
  -c "$src/chain.hip" -o "$root/build/variants/chain_$name.o" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs "$root/build/variants/chain_$name.o" -o "$root/build/variants/lib_chain_$name.so" && echo "built build/variants/lib_chain_$name.so"
