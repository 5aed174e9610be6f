#!/bin/bash
# LDS counters of the wide verify GEMM: bank conflicts vs active cycles, instruction mix.  TAG=r06 T=256 bash scripts/r6/pmc_lds.sh
root=$(cd "$(dirname "$0")/../.." && pwd); tag=${TAG:-r06}; T=${T:-256}
out=$root/gpurun_out/$tag; mkdir -p "$out"; rm -rf "$out/pmc_lds"
( cd /tmp && export TMPDIR=/tmp && T=$T timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_INSTS_SALU \
   --kernel-trace --output-format csv -d "$out/pmc_lds" -- python "$root/scripts/vgemm_bench.py" - pmc > "$out/pmc_lds.log" 2>&1 )
f=$(find "$out/pmc_lds" -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if "vgemm" not in r["Kernel_Name"] and "verify_gemm" not in r["Kernel_Name"]: continue
    agg[(r["Kernel_Name"].split("(")[0][:40], int(r["Grid_Size"]) // int(r["Workgroup_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    print(k, {n: round(v) for n, v in m.items()}, "conflict/active", round(m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m.get("SQ_LDS_IDX_ACTIVE", 1), 1), 3))
PY
rm -rf "$out/pmc_lds"
