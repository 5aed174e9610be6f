cd /root/repo
timeout 900 python -m pytest tests/test_chain.py tests/test_gemv.py -m gpu -q -x 2>&1 | grep -v "^  File\|WARNING" | tail -6
for rep in 1 2; do for v in 1 0; do
  if [ $v = 0 ]; then export UMB_NO_HEAD_STREAM=1; else unset UMB_NO_HEAD_STREAM; fi
  echo "== streamed head $v (rep $rep)"; T1B=1,3 timeout 200 python scripts/ll_bench.py fwd1b 2>&1 | grep forward | sed "s/meta-llama.Llama-3.2-1B-Instruct//" | cut -c1-60
done; done
unset UMB_NO_HEAD_STREAM
