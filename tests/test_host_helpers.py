"""Host-side helpers of the product (CPU): the reference-surface utilities against the golden vectors recorded from
the reference, the tokenizer stand-in, mask packing, configuration registry plumbing."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLD

OPS = np.load(os.path.join(GOLD, "ops.npz"))


def test_logit_helpers_match_reference_vectors():
    from umbrella_amd.speculation import speculation_utils as su
    lg, ids = torch.from_numpy(OPS["rp_logits"]), torch.from_numpy(OPS["rp_ids"])
    assert torch.equal(su.apply_repetition_penalty(ids, lg, 1.05), torch.from_numpy(OPS["rp_out"]))
    assert torch.equal(su.apply_topk(lg, 8), torch.from_numpy(OPS["topk_out"]))
    m = su.make_causal_mask((1, 5), "cpu")
    assert m.dtype == torch.bool and torch.equal(m, torch.tril(torch.ones(5, 5, dtype=torch.bool)))
    assert su.find_first_element_position(torch.tensor([[7, 3, 9, 5]]), [5, 3]) == 1
    assert su.find_first_element_position(torch.tensor([7, 9]), [5, 3]) == -1


@pytest.mark.parametrize("text,done", [("It works.", True), ("really?  ", True), ("好的。", True), ("wait", False),
                                       ("a.b", False), ("", False), ("Stop!\n", True)])
def test_sentence_complete_rule(text, done):
    from umbrella_amd.speculation.speculation_utils import is_sentence_complete_regex
    assert is_sentence_complete_regex(text) is done            # speculation_utils.py:356-358


def test_id_tokenizer_roundtrip():
    from umbrella_amd.speculation.speculation_utils import IdTokenizer
    tok = IdTokenizer()
    ids = tok.encode("12 7 99", return_tensors="pt")
    assert ids.tolist() == [[0, 12, 7, 99]]                    # BOS prepended, dropped by append() (static:140)
    assert tok.decode(ids[0, 1:]) == "12 7 99" and tok.encode("5") == [0, 5]


def test_pack_mask_bits_layout():
    from umbrella_amd.models.llama import pack_mask_bits
    m = torch.zeros(3, 130, dtype=torch.bool)
    m[0, 0] = m[1, 63] = m[1, 64] = m[2, 129] = True
    bits = pack_mask_bits(m)
    assert bits.shape == (3, 3) and bits.dtype == torch.int64
    u = lambda x: int(x) & (2 ** 64 - 1)
    assert u(bits[0, 0]) == 1 and u(bits[1, 0]) == 1 << 63 and u(bits[1, 1]) == 1 and u(bits[2, 2]) == 1 << 1
    assert int(bits[0, 1]) == 0 and int(bits[2, 0]) == 0


def test_model_registry_contract():
    """AutoModelLM: the three mappings, ValueError for unknown names, cuda_graph wins over offload
    (auto_model.py:165-182) -- without touching the GPU (construction is lazy until alloc())."""
    from umbrella_amd.models import AutoModelLM
    from umbrella_amd.models.config import KNOWN
    assert set(AutoModelLM._MODEL_MAPPING) == set(KNOWN) == set(AutoModelLM._OFFLOAD_MODEL_MAPPING)
    with pytest.raises(ValueError):
        AutoModelLM.from_pretrained("nobody/unknown-model")
    m = AutoModelLM.from_pretrained("meta-llama/Llama-3.2-1B-Instruct", offload=True, cuda_graph=True, max_length=100)
    assert m.cuda_graph and not m.offload and m.max_length == 128          # rounded up to whole 32-key tiles
    q = AutoModelLM.from_pretrained("Qwen/Qwen2.5-0.5B-Instruct", max_length=64)
    assert q.config.attention_bias and not q.fused


def test_linear_plan_is_a_function_of_shape_only():
    """Split / wave plan of the skinny GEMM depends on (N, K, format) alone -- the basis of batch invariance."""
    import ctypes as C
    from umbrella_amd import _lib
    lib = _lib.load()
    seen = {}
    for (N, K, awq) in [(57344, 8192, 1), (10240, 8192, 1), (8192, 28672, 1), (128256, 2048, 0), (3072, 2048, 0)]:
        for _ in range(2):
            R, S = C.c_int(0), C.c_int(0)
            lib.umb_gemm_plan(N, K, awq, 0, C.byref(R), C.byref(S))
            assert seen.setdefault((N, K, awq), (R.value, S.value)) == (R.value, S.value)
            assert R.value in (1, 2) and 1 <= S.value <= 16 and (K // 128) // S.value >= (4 if awq else 2)


def test_wide_forward_split_rule():
    """umb_gemm_wide_split: the plan's S up to 64 tokens (batch invariance), beyond that the S <= plan that fills
    the verify kernel's 512 block slots once -- the measured optima of the 70B layer shapes."""
    from umbrella_amd import _lib
    lib = _lib.load()
    plan = {"qkv": (10240, 7), "o": (8192, 8), "down": (8192, 8), "gu": (57344, 1)}
    for T in (1, 13, 64):
        for N, S in plan.values():
            assert lib.umb_gemm_wide_split(T, N, S) == S
    want = {257: {"qkv": 6, "o": 8, "down": 8}, 769: {"qkv": 2, "o": 2, "down": 2}, 1024: {"qkv": 3, "o": 2, "down": 2},
            128: {"qkv": 7, "o": 8, "down": 8}}
    for T, row in want.items():
        for name, s in row.items():
            N, S = plan[name]
            assert lib.umb_gemm_wide_split(T, N, S) == s, (T, name)
        assert lib.umb_gemm_wide_split(T, *plan["gu"]) == 1               # SiLU epilogue: never split
    for T in range(65, 1100, 37):                                          # always within [1, plan]
        for N, S in plan.values():
            assert 1 <= lib.umb_gemm_wide_split(T, N, S) <= S


def test_budget_tree_generator_and_shipped_mi355x_tree():
    """generate_budget_tree: level-major numbering the engines rely on (children of a level laid out by parent order,
    rank), monotone expected accept length in the node budget, the budget respected, and the shipped MI355X-tuned
    growmap (scripts/tune_growmap.py) is its output for the reference's default acceptance vector."""
    import json
    import os
    from umbrella_amd.sequoia_utils import DEFAULT_ACC, expected_accept_length, generate_budget_tree, generate_sequoia_tree
    prev = 0.0
    for T in (2, 5, 13, 16, 31, 64):
        gm = generate_budget_tree(T, 6, DEFAULT_ACC)
        assert gm["size"] == T == len(gm["Successors"]) == len(gm["depth"]) == len(gm["mask"])
        cur = 1
        for lv, ids in enumerate(gm["roots"]):
            assert ids == list(range(ids[0], ids[0] + len(ids)))
            for j, i in enumerate(ids):
                b = gm["branches"][lv][j]
                assert gm["Successors"][i] == list(range(cur, cur + b)) and gm["depth"][i] == lv
                assert b <= len(DEFAULT_ACC)
                cur += b
        assert cur == T and max(gm["depth"]) <= 6
        e = expected_accept_length(gm, DEFAULT_ACC)
        assert e > prev
        prev = e
    # for the same budget and depth a budget tree is at least as good as the fixed-width Sequoia tree
    assert expected_accept_length(generate_budget_tree(13, 4, DEFAULT_ACC), DEFAULT_ACC) >= \
        expected_accept_length(generate_sequoia_tree(3, 4), DEFAULT_ACC)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "umbrella_amd", "trees", "mi355x_70b_awq_1b-T16d3.json")) as f:
        assert json.load(f) == generate_budget_tree(16, 3, DEFAULT_ACC)


def test_spec_bench_aggregation():
    """examples/spec_bench.py keeps the reference's loop order and sums (reference examples/spec_bench.py:96-134):
    prefill -> decode -> append -> decode -> reset per prompt; Avg Accept Tokens = sum(tokens) / sum(target steps) and
    TPOT = 1000 * sum(seconds) / sum(tokens), per category and overall -- NOT means of per-prompt ratios."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "spec_bench_example", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "spec_bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)

    class FakeEngine:
        def __init__(self, results):
            self.results, self.calls = list(results), []

        def _prefill(self, ids):
            self.calls.append(("prefill", int(ids.numel())))
            return True

        def _append(self, ids):
            self.calls.append(("append", int(ids.numel())))
            return int(ids.numel()) != 999                      # a turn that does not fit ends the prompt

        def speculative_decoding(self, max_new_tokens):
            self.calls.append(("decode", max_new_tokens))
            return self.results.pop(0)

        def reset(self):
            self.calls.append(("reset",))

    import torch
    t = lambda n: torch.zeros(1, n, dtype=torch.long)
    res = [(100, 0.5, 30), (40, 0.25, 16), (64, 0.4, 16), (7, 0.1, 7)]
    eng = FakeEngine(res)
    prompts = [("writing", [t(64), t(32)]), ("math", [t(128), t(999)]), ("writing", [t(256)])]
    per, total = mod.run_prompts(eng, prompts, gen_len=77)
    assert eng.calls == [("prefill", 64), ("decode", 77), ("append", 32), ("decode", 77), ("reset",),
                         ("prefill", 128), ("decode", 77), ("append", 999), ("reset",),
                         ("prefill", 256), ("decode", 77), ("reset",)]
    assert per["writing"] == [147, 0.85, 53] and per["math"] == [64, 0.4, 16] and total == [211, 1.25, 69]
    rows = mod.report(per, total)
    assert rows[0] == "math | Avg Accept Tokens 4.00 | TPOT 6.25 ms"
    assert rows[1] == "writing | Avg Accept Tokens {:.2f} | TPOT {:.2f} ms".format(147 / 53, 1000 * 0.85 / 147)
    assert rows[2].startswith("Summary | Avg Accept Tokens {:.2f} | TPOT {:.2f} ms".format(211 / 69, 1000 * 1.25 / 211))
