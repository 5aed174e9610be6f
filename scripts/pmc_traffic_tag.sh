#!/bin/bash
# HBM traffic of the dominant kernel (70B gate/up int4 GEMM, T = 13) from separate rocprofv3 --pmc FETCH_SIZE and
# --pmc WRITE_SIZE passes (MI355X_MICROARCH.md HBM section: FETCH_SIZE [KB] x 1024 x 2 on gfx950, WRITE_SIZE [KB] x 1024).
# Writes gpurun_out/$TAG_pmc_gemm70b_traffic.json (TAG default r04).  Run on the GPU box: TAG=r04 bash scripts/pmc_traffic_tag.sh
root=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  out=$root/gpurun_out/pmc_traffic_$c
  rm -rf "$out"; mkdir -p "$out"
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$out" -- python "$root/scripts/gemm_bench.py" 70b > "$out/run.log" 2>&1
done
python - "$root" <<'PY'
import csv, glob, hashlib, json, os, sys
root = sys.argv[1]
raw = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(os.path.join(root, "gpurun_out", f"pmc_traffic_{c}", "**", "*counter_collection.csv"), recursive=True)[0]
    agg = {}
    for r in csv.DictReader(open(f)):
        if "skinny_gemm_kernel" not in r["Kernel_Name"] or r["Counter_Name"] != c:
            continue
        agg.setdefault(f'{r["Grid_Size"]}x{r["Workgroup_Size"]}', []).append(float(r["Counter_Value"]))      # threads x block size
    raw[c] = {str(g): {"launches": len(v), "mean_KB": sum(v) / len(v)} for g, v in agg.items()}
g = f"{256 * 512}x512"
traffic = raw["FETCH_SIZE"][g]["mean_KB"] * 1024 * 2 + raw["WRITE_SIZE"][g]["mean_KB"] * 1024
src = open(os.path.join(root, "umbrella_amd", "csrc", "gemm.hip"), "rb").read()
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) -- python scripts/gemm_bench.py 70b (scripts/pmc_traffic_tag.sh)",
       "kernel": "skinny_gemm_kernel<F16, int4 (folded dequant), TT=1, R=2, 8 waves> gate_up N=57344 K=8192 T=13 (grid 256 x 512: one 14-tile block per CU)",
       "correction": "FETCH_SIZE [KB] x 1024 x 2 (gfx950 reports half the bytes of wide coalesced streaming reads); WRITE_SIZE [KB] x 1024",
       "raw": raw, "gate_up_traffic_bytes": traffic, "gate_up_algorithmic_bytes": 249561088,
       "gemm_hip_sha256_16": hashlib.sha256(src).hexdigest()[:16]}
json.dump(out, open(os.path.join(root, "gpurun_out", os.environ.get("TAG", "r04") + "_pmc_gemm70b_traffic.json"), "w"), indent=1)
print(json.dumps({k: out[k] for k in ("gate_up_traffic_bytes", "gate_up_algorithmic_bytes", "gemm_hip_sha256_16")}))
PY
find "$root/gpurun_out" -name "*.csv" -size +20M -delete
