#!/bin/bash
root=$(cd "$(dirname "$0")/../.." && pwd); cd "$root"
for m in "tail 3" "full 3"; do echo "=== $m"; timeout 120 python scripts/r5/chain_debug.py $m 2>&1 | grep -E "mismatch|tokens" | grep -v "mismatches 0 /"; done
echo "=== bf16 full 3"; DT=bf16 timeout 120 python scripts/r5/chain_debug.py full 3 2>&1 | grep -E "mismatch|tokens" | grep -v "mismatches 0 /"
echo "=== bf16 front 3"; DT=bf16 timeout 120 python scripts/r5/chain_debug.py front 3 2>&1 | grep -E "mismatch|tokens" | grep -v "mismatches 0 /"
timeout 600 python -m pytest tests/test_chain.py -q 2>&1 | grep -v "^  File" | tail -5
for T in 3; do UMB_LIB_PATH=build/variants/lib_chain_trace.so timeout 300 python scripts/r5/chain_trace.py $T 8 2>&1 | grep -v "WARNING\|amdgpu.ids"; done
