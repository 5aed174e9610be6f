#!/bin/bash
# rocprofv3 kernel + memory-copy trace of the layer-streamed dynamic engine (BASELINE config 3): shows the host->device
# slab copies of the next verify running while the draft tree's kernels execute.  Summary -> gpurun_out/prof_offload_overlap.json
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/prof_offload
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$out" -- python "$root/scripts/bench_configs.py" --config c3 --steps 3 > "$out/run.log" 2>&1
python - "$out" "$root" <<'PY'
import csv, glob, json, os, sys
out, root = sys.argv[1], sys.argv[2]
kt = glob.glob(os.path.join(out, "**", "*kernel_trace.csv"), recursive=True)[0]
mt = glob.glob(os.path.join(out, "**", "*memory_copy_trace.csv"), recursive=True)[0]
copies = []
for r in csv.DictReader(open(mt)):
    b = int(r.get("Bytes", r.get("bytes", 0)) or 0)
    if b >= 100e6 and "HOST_TO_DEVICE" in (r.get("Direction", "") + r.get("Name", "")).upper().replace(" ", "_"):
        copies.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), b))
kern = []
for r in csv.DictReader(open(kt)):
    n = r["Kernel_Name"]
    kern.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n[:40], int(r.get("Grid_Size_X", r.get("Grid_Size", 0)))))
copies.sort(); kern.sort()
# draft kernels = ll_gemm / skinny kernels with the 1B draft's small grids + topk / beam kernels between verifies
def overlap(a0, a1, b0, b1): return max(0, min(a1, b1) - max(a0, b0))
draft_names = ("topk_rows", "beam_expand")
marks = [k for k in kern if any(d in k[2] for d in draft_names)]
tot_copy = sum(c[1] - c[0] for c in copies)
ov = 0
for c in copies:
    for k in marks:
        if k[0] > c[1]: break
        ov += overlap(c[0], c[1], k[0], k[1])
first, last = copies[0][0], copies[-1][1]
res = {"h2d_slab_copies": len(copies), "slab_bytes": copies[0][2] if copies else 0,
       "copy_busy_ms": round(tot_copy / 1e6, 2), "span_ms": round((last - first) / 1e6, 2),
       "draft_marker_kernels": len(marks),
       "copies_with_draft_kernel_inside": sum(1 for c in copies if any(c[0] <= k[0] <= c[1] for k in marks)),
       "note": "a slab copy 'has a draft kernel inside' when a top-k / beam-expand kernel of the draft tree starts while the copy is in flight"}
json.dump(res, open(os.path.join(root, "gpurun_out", "prof_offload_overlap.json"), "w"), indent=1)
print(json.dumps(res))
PY
tail -2 "$out/run.log"
find "$out" -name "*.csv" -size +20M -delete
