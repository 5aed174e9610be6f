#!/bin/bash
# What a loader-wave (LDS-DMA) stream of the 70B layer GEMMs' footprints costs PER LAUNCH (kernel duration on the profiler's clock, ramp
# included): 9 / 11 / 30 / 61 slots of 16 KiB per CU = o 35.7 / qkv 44.6 / down 125 / gate-up 250 MB.  The floor of an engine-style GEMM.
root=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p "$root/gpurun_out"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 "$root/scripts/probe/ldsdma_probe.hip" -o "$root/gpurun_out/ldsdma_probe" || exit 1
cd /tmp && export TMPDIR=/tmp
for s in 9 11 30 61; do
  out=$root/gpurun_out/prof_fill; rm -rf "$out"; mkdir -p "$out"
  ONLY=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- "$root/gpurun_out/ldsdma_probe" $s > "$out/run.log" 2>&1
  f=$(find "$out" -name "*kernel_stats.csv" | head -1)
  python - "$f" $s <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "fill_kernel" in r["Name"]:
        s = int(sys.argv[2]); mb = 256 * s * 16384 / 1e6
        print(f"{s:3d} slots per CU ({mb:6.1f} MB): kernel avg {float(r['AverageNs'])/1e3:6.2f} us  min {float(r['MinNs'])/1e3:6.2f}  -> {mb/float(r['AverageNs'])*1e3/1e3:5.2f} TB/s;  eager back-to-back: {open(sys.argv[1].rsplit('/',2)[0]+'/run.log').read().strip().splitlines()[-1][-40:]}")
PY
done
