"""Per-kernel counter table of one graph-replayed forward (scripts/r5/pmc_table.sh).

  python scripts/r5/pmc_table.py <base dir with trace/ fetch/ write/ sq/> <out.csv> <fwd70b|fwd1b|fwd8bawq>

Columns: kernel, workgroups, calls, avg_us (kernel-trace pass, no counters), algorithmic MB (the linear's weight bytes incl.
AWQ tile metadata; blank for kernels that move activations only), traffic MB = FETCH_SIZE [KB] x 1024 x 2 + WRITE_SIZE [KB] x 1024
(MI355X_MICROARCH.md, HBM: gfx950 reports half of a wide coalesced streaming read), traffic / algorithmic, achieved TB/s on the
algorithmic bytes and its fraction of the 8 TB/s peak, and the SQ wave-cycle split (disjoint: active / issue-stalled / parked)."""
import collections
import csv
import glob
import os
import sys

base, dst, fwd = sys.argv[1], sys.argv[2], sys.argv[3]
PEAK = 8.0e12


def awq_bytes(N, K):
    return N * K // 2 + (N // 16) * (K // 128) * 64


def dense_bytes(N, K):
    return N * K * 2


# (kernel-name substring, workgroups) -> (label, algorithmic bytes); shapes of the three forwards
ALG = {
    "fwd70b": {  # Llama-3.1-70B AWQ: H 8192, I 28672, q/k/v N 10240
        ("skinny_gemm_kernel", 560): ("qkv int4 N10240 K8192", awq_bytes(10240, 8192)),
        ("skinny_gemm_kernel", 256): ("o / gate-up int4 (see template)", None),
        ("skinny_gemm_kernel", 512): ("down int4 N8192 K28672", awq_bytes(8192, 28672)),
        ("skinny_gemm_kernel", 1002): ("lm_head fp16 N128256 K8192", dense_bytes(128256, 8192)),
    },
    "fwd1b": {  # Llama-3.2-1B fp16: H 2048, I 8192, q/k/v N 3072
        ("gv_kernel<F16, 2, 1, 2>", 256): ("gate/up fp16 N16384 K2048", dense_bytes(16384, 2048)),
        ("gv_kernel<F16, 4, 4, 4>", 256): ("down fp16 N2048 K8192", dense_bytes(2048, 8192)),
        ("gv_kernel<F16, 2, 1, 3>", 256): ("qkv fp16 N3072 K2048", dense_bytes(3072, 2048)),
        ("gv_kernel<F16, 1, 1, 4>", 256): ("o fp16 N2048 K2048", dense_bytes(2048, 2048)),
        ("skinny_gemm_kernel", 1002): ("lm_head fp16 N128256 K2048", dense_bytes(128256, 2048)),
        ("draft_head_kernel", 256): ("engine lm_head fp16 N128256 K2048 (tied table, row-major)", dense_bytes(128256, 2048)),
        ("draft_chain_kernel", 256): ("engine chain o + gate/up + down + qkv (1 launch)", dense_bytes(2048, 2048) + dense_bytes(16384, 2048) + dense_bytes(2048, 8192) + dense_bytes(3072, 2048)),
    },
    "fwd8bawq": {  # Llama-3.1-8B AWQ: H 4096, I 14336, q/k/v N 6144
        ("skinny_gemm_kernel", 768): ("qkv int4 N6144 K4096", awq_bytes(6144, 4096)),
        ("skinny_gemm_kernel", 512): ("down int4 N4096 K14336", awq_bytes(4096, 14336)),
        ("skinny_gemm_kernel", 1002): ("lm_head fp16 N128256 K4096", dense_bytes(128256, 4096)),
    },
}
# template-disambiguated entries (o and gate/up of the 70B both launch 256 workgroups; gate/up is the 8-wave form)
ALG_T = {
    "fwd70b": {("skinny_gemm_kernel<F16, 1, 1, 2, 2, 8>", 256): ("gate/up int4 N57344 K8192", awq_bytes(57344, 8192)),
               ("skinny_gemm_kernel<F16, 1, 1, 2, 2, 4>", 256): ("o int4 N8192 K8192", awq_bytes(8192, 8192))},
    "fwd8bawq": {("2, 8>", 256): ("gate/up int4 N28672 K4096", awq_bytes(28672, 4096)),
                 ("2, 4>", 256): ("o int4 N4096 K4096", awq_bytes(4096, 4096))},
}


def find(sub, pat):
    f = glob.glob(os.path.join(base, sub, "**", pat), recursive=True)
    return f[0] if f else None


def key_of(r):
    name = r["Kernel_Name"].split("(")[0][:90]
    if "Grid_Size_X" in r:
        blocks = 1
        for d in "XYZ":
            blocks *= max(1, int(r.get(f"Grid_Size_{d}", 1) or 1)) // max(1, int(r.get(f"Workgroup_Size_{d}", 1) or 1))
    else:
        blocks = int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)
    return name, blocks


dur = collections.defaultdict(list)
t = find("trace", "*kernel_trace.csv")
if t:
    for r in csv.DictReader(open(t)):
        dur[key_of(r)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
ctr = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("fetch", "write", "sq"):
    f = find(sub, "*counter_collection.csv")
    if not f:
        continue
    for r in csv.DictReader(open(f)):
        ctr[key_of(r)][r["Counter_Name"]].append(float(r["Counter_Value"]))


def mean(v):
    return sum(v) / len(v) if v else None


def alg_of(name, blocks):
    for (sub, b), v in ALG_T.get(fwd, {}).items():
        if sub in name and b == blocks:
            return v
    for (sub, b), v in ALG.get(fwd, {}).items():
        if sub in name and b == blocks and v[1] is not None:
            return v
    return ("", None)


keep = ("skinny_gemm", "gv_kernel", "ll_gemm", "reduce_", "tree_attn", "embed", "rmsnorm", "draft_chain", "draft_head", "topk", "verify_gemm")
rows = []
for k, v in dur.items():
    name, blocks = k
    if not any(s in name for s in keep) or len(v) < 8:
        continue
    us = sum(v) / len(v) / 1e3
    c = ctr.get(k, {})
    fetch, write = mean(c.get("FETCH_SIZE", [])), mean(c.get("WRITE_SIZE", []))
    traffic = (fetch * 1024 * 2 if fetch is not None else 0) + (write * 1024 if write is not None else 0) if (fetch is not None or write is not None) else None
    label, alg = alg_of(name, blocks)
    wc = mean(c.get("SQ_WAVE_CYCLES", []))
    split = ["", "", ""]
    if wc:
        split = [round((mean(c.get(n, [])) or 0) / wc, 3) for n in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY")]
    rows.append([name, blocks, len(v), round(us, 2), label,
                 round(alg / 1e6, 2) if alg else "", round(traffic / 1e6, 2) if traffic is not None else "",
                 round(traffic / alg, 3) if (alg and traffic is not None) else "",
                 round(alg / us / 1e6, 3) if alg else "", round(alg / (us * 1e-6) / PEAK, 3) if alg else "",
                 *split, round(sum(v) / 1e6, 3)])
rows.sort(key=lambda r: -r[-1])
with open(dst, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "workgroups", "calls", "avg_us", "what", "algorithmic_MB", "traffic_MB", "traffic_over_algorithmic",
                "TBps_on_algorithmic", "frac_of_8TBps", "wave_active", "wave_issue_stalled", "wave_parked", "total_ms"])
    w.writerows(rows)
print("wrote", dst, len(rows), "rows")
for r in rows[:14]:
    print(r)
