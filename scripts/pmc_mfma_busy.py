"""rocprofv3 --pmc counter_collection.csv -> JSON: per (kernel, grid) means of the SQ counters and the MFMA-busy fraction
of SIMD time,  SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)  (MI355X_MICROARCH.md: the counter
counts SIMD cycles -- 16 per v_mfma_f32_16x16x32_f16, 32 per 32x32x16; GRBM_GUI_ACTIVE is summed over the XCDs).
  python scripts/pmc_mfma_busy.py <counter_collection.csv> <out.json> "<command>" [name-substring ...]"""
import collections
import csv
import json
import sys

src, dst, cmd, pats = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4:]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
with open(src) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        if pats and not any(p in name for p in pats):
            continue
        key = (name.split("(")[0][:80], int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
import hashlib, os
_csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "umbrella_amd", "csrc")
_h = lambda f: hashlib.sha256(open(os.path.join(_csrc, f), "rb").read()).hexdigest()[:16]
out = {"command": cmd, "gemm_hip_sha256_16": _h("gemm.hip"), "vgemm_hip_sha256_16": _h("vgemm.hip"), "notes": "means per launch; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs); "
       "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles summed over waves", "kernels": []}
for (name, blocks), ctr in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("GRBM_GUI_ACTIVE", [0]))):
    m = {c: sum(v) / len(v) for c, v in ctr.items()}
    row = {"kernel": name, "blocks": blocks, "launches": max(len(v) for v in ctr.values()), **{c: round(x, 1) for c, x in sorted(m.items())}}
    if m.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        row["mfma_busy"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8 * 1024), 4)
    if m.get("SQ_WAVE_CYCLES"):
        w = m["SQ_WAVE_CYCLES"]
        row["wave_cycle_split"] = {k: round(m[c] / w, 3) for k, c in (("active", "SQ_ACTIVE_INST_ANY"), ("issue_stall", "SQ_WAIT_INST_ANY"),
                                                                       ("parked_waitcnt_barrier", "SQ_WAIT_ANY"), ("lds_issue_stall", "SQ_WAIT_INST_LDS")) if c in m}
    out["kernels"].append(row)
with open(dst, "w") as f:
    json.dump(out, f, indent=1)
for r in out["kernels"][:12]:
    print(r["kernel"][:60], r["blocks"], "mfma_busy", r.get("mfma_busy"), r.get("wave_cycle_split"))
