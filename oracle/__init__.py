"""CPU oracle for the UMbreLLa speculative-decoding hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain torch-CPU / numpy
restatement of the reference algorithm (Infini-AI-Lab/UMbreLLa @ 2025-02-16)
for the draft-expand / verify path.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it -- and there only as the
checker, never as the thing measured or shipped.  The product
(``umbrella_amd``) never imports ``oracle`` and raises when its HIP library is
missing.

Pinning status
--------------
* Engines / model runtime / KV semantics / accept scan / Sequoia generator:
  PINNED.  ``tests/golden/make_golden.py`` imports the reference itself (CPU
  shims for the absent third-party wheels) and records golden vectors that
  ``tests/test_oracle_golden.py`` replays against this restatement.
* ``flashinfer`` (unpinned wheel, install.sh:2) and ``autoawq-kernels==0.0.8``
  (requirements.txt:7) arithmetic is NOT in /root/reference.  Their published
  algorithms are restated here (RMSNorm fp32-accumulate; masked softmax
  attention == umbrella/attn/cache.py:169-192; AutoAWQ GEMM packing order
  [0,2,4,6,1,3,5,7], W=(q-z)*s, group 128).  For those kernels: parity
  unpinned beyond the reference's own call sites and the HF LlamaForCausalLM
  forward (which make_golden checks against).
"""
