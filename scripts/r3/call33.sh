#!/bin/bash
cd $GRAFT_REPO_ROOT
o=gpurun_out/c33; mkdir -p $o; rm -f $o/*.log
timeout 2000 python -m pytest tests -m gpu -q -x > $o/tests.log 2>&1; echo "tests rc=$?" >> $o/tests.log
tail -n 3 $o/tests.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $o/bench.json 2> $o/bench.err
UMB_VG_PP=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $o/bench_nopp.json 2> $o/bench_nopp.err
python - <<'PY'
import json
for f in ("gpurun_out/c33/bench.json", "gpurun_out/c33/bench_nopp.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["ms_per_step"], {k: {kk: v.get(kk) for kk in ("ms_per_step", "verify_TFLOPs", "tokens_per_s", "prefill_tokens_per_s") if isinstance(v, dict) and kk in v} for k, v in d.get("secondary", {}).items()})
PY
