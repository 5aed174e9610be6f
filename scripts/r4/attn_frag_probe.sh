for lib in base fragprobe; do
  echo "== $lib"
  export UMB_LIB_PATH=build/variants/lib_$lib.so
  for p in 100 250 500 760 1000; do python scripts/attn_bench.py --T 13 --prefix $p --Lmax 2048 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('T13', d['prefix'], d['us_per_launch(attn+combine)'])"; done
  python scripts/attn_bench.py --T 32 --prefix 900 --Lmax 4096 --Hq 32 --Hkv 8 --D 128 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('T32 8B', d['prefix'], d['us_per_launch(attn+combine)'])"
  python scripts/attn_bench.py --T 3 --prefix 400 --Lmax 2048 --Hq 32 --Hkv 8 --D 64 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('T3 1B', d['prefix'], d['us_per_launch(attn+combine)'])"
  for T in 257 769; do python scripts/attn_bench.py --T $T --prefix 128 --Lmax 4096 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wide', d['T'], d['us_per_launch(attn+combine)'])"; done
  python scripts/attn_bench.py --T 1024 --prefix 1024 --Lmax 4096 --causal 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('causal1024@1024', d['us_per_launch(attn+combine)'])"
done
