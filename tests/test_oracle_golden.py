"""Pin the CPU oracle against vectors recorded from the reference itself
(tests/golden/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import replay_case, GOLD, load_golden, oracle_engine_for_case, oracle_model
from oracle import ops, sequoia
from oracle.model import AppendKV, SlotKV

G = load_golden()
OPS = np.load(os.path.join(GOLD, "ops.npz"))
T = torch.from_numpy


def test_reference_matches_hf_forward():
    assert G["hf_vs_ref_max_abs"] < 1e-4
    assert G["sequoia_3x4_equals_shipped"] is True


def test_rope_matches_reference():
    qe, ke = ops.apply_rope(T(OPS["rope_q"]), T(OPS["rope_k"]), T(OPS["rope_cos"]), T(OPS["rope_sin"]),
                            T(OPS["rope_pos"]))
    assert torch.equal(qe, T(OPS["rope_qe"])) and torch.equal(ke, T(OPS["rope_ke"]))


def test_rope_tables_match_hf_inv_freq():
    """Both frequency tables -- the product's (models/config.py) and the oracle's own (oracle/ops.py; what tests/helpers.py
    builds the oracle model from) -- against the values recorded from HF's rotary module, and against each other on the
    real configurations incl. the llama3 frequency scaling of the 1B / 8B / 70B models."""
    from umbrella_amd.models.config import KNOWN, LlamaCfg, rope_inv_freq
    cfg = LlamaCfg(**G["target_cfg"])
    ml = np.load(os.path.join(GOLD, "model_logits.npz"))
    inv, scale = rope_inv_freq(cfg)
    oinv, oscale = ops.rope_inv_freq(cfg.head_dim, cfg.rope_theta, cfg.rope_scaling)
    assert scale == 1.0 and oscale == 1.0
    np.testing.assert_allclose(inv.numpy(), ml["inv_freq"], rtol=1e-6)
    np.testing.assert_allclose(oinv.numpy(), ml["inv_freq"], rtol=1e-6)
    for name in ("meta-llama/Llama-3.2-1B-Instruct", "meta-llama/Llama-3.1-8B-Instruct",
                 "hugging-quants/Meta-Llama-3.1-70B-Instruct-AWQ-INT4"):
        c = KNOWN[name]
        a, _ = rope_inv_freq(c)
        b, _ = ops.rope_inv_freq(c.head_dim, c.rope_theta, c.rope_scaling)
        np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=2e-6)
        assert c.rope_scaling and float(a[-1]) < float(a[0])


def test_masked_attention_matches_reference_static_cache():
    # StaticKV_Cache.compute_attention (cache.py:169-192) is HND; oracle is NHD
    kc, vc = T(OPS["attn_kcache"]).clone(), T(OPS["attn_vcache"]).clone()     # [Hkv, Lmax, D]
    sids = torch.arange(9, 16)
    kc[:, sids] = T(OPS["attn_knew"]); vc[:, sids] = T(OPS["attn_vnew"])
    q = T(OPS["attn_q"]).permute(1, 0, 2)                                       # [T, Hq, D]
    out = ops.masked_attention(q, kc.permute(1, 0, 2), vc.permute(1, 0, 2), T(OPS["attn_mask"]))
    torch.testing.assert_close(out, T(OPS["attn_out"]), rtol=1e-5, atol=1e-6)


def test_kv_gather_matches_reference():
    for cls in (AppendKV, SlotKV):
        c = cls(1, 32, 2, 64, torch.float32)
        c.k.copy_(T(OPS["gather_k_before"])); c.v.copy_(T(OPS["gather_v_before"]))
        c.gather_kv_incremental(torch.tensor([9, 11, 14]), 9)
        assert torch.equal(c.k, T(OPS["gather_k_after"])) and torch.equal(c.v, T(OPS["gather_v_after"]))
        assert c.kv_offset == int(OPS["gather_offset"])


def test_fragment_ordered_cache_speaks_the_reference_gather():
    """The product's KV cache stores each (layer, kv head) slab in MFMA fragment order (umbrella_amd/attn/cache.py); its
    host-side views must still behave like the reference's cache: the recorded `gather_kv_incremental` vector of the reference
    (KV_Cache, cache.py:41-49; [L, Lmax, Hkv, D]) through set_rows -> gather -> k_rows / v_rows, the converters round trip, and the
    offset formulas are permutations of a slab.  CPU only (torch index arithmetic; the device kernels share the formulas)."""
    from umbrella_amd.attn.cache import (TreeKVCache, VT_PAD, k_from_frag, k_offsets, k_to_frag, vt_from_frag, vt_offsets,
                                         vt_to_frag)
    kb, vb = T(OPS["gather_k_before"]), T(OPS["gather_v_before"])            # [1, Lmax 32, Hkv 2, D 64]
    L, Lmax, Hkv, D = kb.shape
    c = TreeKVCache(L, Hkv, D, Lmax, "cpu", torch.float32)
    allpos = torch.arange(Lmax)
    c.set_rows(allpos, kb.permute(0, 2, 1, 3), vb.permute(0, 2, 1, 3))
    c.gather_kv_incremental(torch.tensor([9, 11, 14]), 9)
    assert c.kv_offset == int(OPS["gather_offset"])
    n = c.kv_offset                       # the reference zeroes the tail (cache.py:46-47); this cache leaves it, it is never visible
    assert torch.equal(c.k_rows(allpos[:n]).permute(0, 2, 1, 3), T(OPS["gather_k_after"])[:, :n])
    assert torch.equal(c.v_rows(allpos[:n]).permute(0, 2, 1, 3), T(OPS["gather_v_after"])[:, :n])
    # storage <-> semantic converters
    ks = k_from_frag(c.k)
    assert torch.equal(ks[:, :, :n], T(OPS["gather_k_after"]).permute(0, 2, 1, 3)[:, :, :n]) and torch.equal(k_to_frag(ks), c.k)
    vs = vt_from_frag(c.vt)
    assert torch.equal(vs[..., :n], T(OPS["gather_v_after"]).permute(0, 2, 3, 1)[..., :n]) and torch.equal(vt_to_frag(vs), c.vt)
    assert float(c.vt.reshape(L, Hkv, -1)[..., Lmax * D:].abs().max()) == 0.0 and c.vt.shape[-1] == Lmax + VT_PAD
    for Dh, Lm in ((32, 64), (64, 96), (128, 160)):
        ko, vo = k_offsets(torch.arange(Lm), Dh).reshape(-1), vt_offsets(torch.arange(Lm), Dh).reshape(-1)
        assert torch.equal(ko.sort().values, torch.arange(Lm * Dh)) and torch.equal(vo.sort().values, torch.arange(Lm * Dh))
        # one K fragment = 16 keys x 32 features in 512 consecutive elements; one V^T fragment = 16 features x 32 keys
        frag = (ko.reshape(Lm, Dh)[:32, :32] // 512).unique()
        assert frag.numel() == 2 and (vo.reshape(Dh, Lm)[:16, :32] // 512).unique().numel() == 1


def test_logit_helpers_match_reference():
    lg, ids = T(OPS["rp_logits"]), T(OPS["rp_ids"])
    assert torch.equal(ops.repetition_penalty(ids, lg, 1.05), T(OPS["rp_out"]))
    assert torch.equal(ops.keep_topk(lg, 8), T(OPS["topk_out"]))
    assert torch.equal(ops.topk_flatten_gather(lg[:3], 2, torch.tensor([0, 1, 2, 4])), T(OPS["argmax_gather"]))


def test_sequoia_generator_kat():
    with open(os.path.join(GOLD, "growmaps.json")) as f:
        gm = json.load(f)
    assert sequoia.generate(3, 4) == gm["3x4"]
    assert sequoia.generate(5, 6, gm["5x6_acc"]) == gm["5x6"]


def test_shipped_growmaps_are_generator_outputs():
    """umbrella_amd/trees/*.json: each file is what this repository's generator produces for the recorded acceptance
    vector (scripts/fit_growmaps.py found vectors whose trees have the topology of the growmaps the reference
    ships under the same names), and satisfies the layout the static engine relies on."""
    from umbrella_amd.sequoia_utils import generate_sequoia_tree, successor_list_to_mask
    tdir = os.path.join(os.path.dirname(GOLD), "..", "umbrella_amd", "trees")
    with open(os.path.join(tdir, "acceptance_vectors.json")) as f:
        vectors = json.load(f)
    assert {"sequoia_tree-3x4.json", "sequoia_tree-5x6.json", "8b_sequoia_tree-5x6.json"} <= set(vectors)
    for name, acc in vectors.items():
        with open(os.path.join(tdir, name)) as f:
            gm = json.load(f)
        w, d = len(gm["roots"][1]), len(gm["roots"]) - 1
        assert gm == generate_sequoia_tree(w, d, acc=acc), name
        assert gm["size"] == w * d + 1 and gm["mask"] == successor_list_to_mask(gm["Successors"])
        for lvl, ids in enumerate(gm["roots"]):
            assert ids == list(range(ids[0], ids[0] + len(ids)))
            kids = [c for v in ids for c in gm["Successors"][v]]
            if lvl + 1 < len(gm["roots"]):
                assert kids == gm["roots"][lvl + 1] and [len(gm["Successors"][v]) for v in ids] == gm["branches"][lvl]
    with open(os.path.join(GOLD, "growmaps.json")) as f:
        ref34 = json.load(f)["3x4"]
    with open(os.path.join(tdir, "sequoia_tree-3x4.json")) as f:
        assert json.load(f) == ref34                      # same tree as the reference's shipped 3x4


def test_all_six_reference_growmaps_are_shipped():
    """Every growmap the reference ships exists here under the same name and is the same data (sha256 of the canonical
    JSON, recorded from /root/reference by scripts/fit_growmaps.py); the one topology no acceptance vector reproduces
    (5x8: a score tie in the original run) is rebuilt from its recorded branch table."""
    import hashlib
    from umbrella_amd.sequoia_utils import growmap_from_branches
    tdir = os.path.join(os.path.dirname(GOLD), "..", "umbrella_amd", "trees")
    with open(os.path.join(GOLD, "ref_tree_digests.json")) as f:
        digests = json.load(f)
    assert len(digests) == 6
    for name, want in digests.items():
        with open(os.path.join(tdir, name)) as f:
            gm = json.load(f)
        assert hashlib.sha256(json.dumps(gm, sort_keys=True, separators=(",", ":")).encode()).hexdigest() == want, name
    with open(os.path.join(tdir, "branch_tables.json")) as f:
        for name, table in json.load(f).items():
            with open(os.path.join(tdir, name)) as g:
                assert json.load(g) == growmap_from_branches(table), name
    # the 3x4 tree from its own branch table: the two constructions agree where both apply
    with open(os.path.join(tdir, "sequoia_tree-3x4.json")) as f:
        gm = json.load(f)
    assert growmap_from_branches(gm["branches"]) == gm


def test_reference_config_growmap_paths_resolve():
    """A reference config's ``growmap_path`` (relative to the reference's examples directory) resolves to the shipped
    tree of the same name; unknown names fail loudly."""
    from umbrella_amd.speculation.static_speculation_engine import resolve_growmap_path
    for name in ("sequoia_tree-3x4.json", "sequoia_tree-5x6.json", "8b_sequoia_tree-6x7.json"):
        p = resolve_growmap_path("../umbrella/trees/" + name)
        assert os.path.basename(p) == name and os.path.exists(p)
    with pytest.raises(FileNotFoundError):
        resolve_growmap_path("../umbrella/trees/no_such_tree.json")


def test_reference_configs_are_consumable():
    """Every config the reference ships (engine keys recorded in tests/golden/ref_config_facts.json by
    make_ref_config_facts.py) is accepted by AutoEngine unchanged: engine kind, registry names (the small code
    drafters need a local directory), growmap path, tree limits.  Construction only -- initialize() needs the GPU."""
    from umbrella_amd.models.config import KNOWN
    from umbrella_amd.speculation.auto_engine import AutoEngine
    from umbrella_amd.speculation.static_speculation_engine import resolve_growmap_path
    with open(os.path.join(GOLD, "ref_config_facts.json")) as f:
        facts = json.load(f)
    assert len(facts) == 11
    local_only = {"InfiniAILab/CodeDrafter-500M"}
    for name, rec in facts.items():
        cfg = rec["engine_kwargs"]
        eng = AutoEngine.from_config("cuda:0", **dict(cfg))
        assert type(eng).__name__ == ("StaticSpeculationEngine" if cfg["engine"] == "static" else "DynamicSpeculationEngine")
        for key in ("model", "draft_model"):
            assert cfg[key] in KNOWN or cfg[key] in local_only, (name, cfg[key])
        if cfg["engine"] == "static":
            assert os.path.exists(resolve_growmap_path(cfg["growmap_path"]))
        else:
            assert cfg["width"] * cfg["num_beams"] <= 1024 and cfg["width"] <= 64
            assert cfg["width"] * cfg["depth"] + 1 <= 1024             # accept scan limit


def test_shipped_configs_load():
    """configs/*.yaml parse into engine kwargs AutoEngine accepts (construction only)."""
    from umbrella_amd.speculation.auto_engine import AutoEngine
    from umbrella_amd.utils import load_config
    cdir = os.path.join(os.path.dirname(GOLD), "..", "configs")
    names = sorted(n for n in os.listdir(cdir) if n.endswith(".yaml"))
    assert len(names) >= 3
    for n in names:
        cfg = load_config(os.path.join(cdir, n))
        for k in ("generation_length", "max_turns", "template"):
            cfg.pop(k, None)
        assert AutoEngine.from_config("cuda:0", **cfg) is not None


def test_model_logits_match_reference():
    ml = np.load(os.path.join(GOLD, "model_logits.npz"))
    m = oracle_model(G["target_cfg"], G["seeds"]["target"], 128)
    ids = T(ml["prompt"])[None]
    n = ids.shape[1]
    mask = torch.tril(torch.ones(n, 128, dtype=torch.bool))
    logits = m.inference(ids, torch.arange(n)[None], mask, torch.arange(n))
    np.testing.assert_allclose(logits[0, -1].numpy(), ml["logits_last"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(logits[0, ::6, :64].numpy(), ml["logits_rows"], rtol=1e-4, atol=1e-4)


def _replay(case_name):
    replay_case(G["cases"][case_name], oracle_engine_for_case(G, case_name), case_name)


@pytest.mark.parametrize("case", sorted(G["cases"].keys()))
def test_engine_replays_reference_trace(case):
    _replay(case)


def test_greedy_spec_equals_greedy_ar():
    """Property oracle: greedy speculative output == greedy AR output of the target."""
    for case, key in (("static_3x4_selfdraft", "hf_greedy_prompt2"), ("static_3x4", "hf_greedy_prompt"),
                      ("dynamic_w8b8d4_selfdraft", "hf_greedy_prompt2"), ("dynamic_w4b6d3", "hf_greedy_prompt")):
        toks = G["cases"][case]["turns"][0]["tokens"]
        ar = G[key]
        n = min(len(toks), len(ar))
        assert toks[:n] == ar[:n], case


def test_accept_scan_synthetic_cases():
    gm = sequoia.generate(3, 4)
    mask = torch.tensor(gm["mask"]) == 1
    want = mask.sum(-1)
    parents = torch.zeros(13, dtype=torch.int32)
    for v, s in enumerate(gm["Successors"]):
        parents[s] = v
    spec = torch.arange(100, 113)
    # all reject: only the root; bonus = sampled[0]
    sampled = torch.full((13,), 7)
    path, bonus = ops.accept_scan(sampled, spec, parents, mask, want)
    assert path.tolist() == [0] and bonus == 7
    # full path 0 -> 1 -> 4 -> 7 -> 10
    sampled = torch.full((13,), 7)
    sampled[0], sampled[1], sampled[4], sampled[7], sampled[10] = 101, 104, 107, 110, 999
    path, bonus = ops.accept_scan(sampled, spec, parents, mask, want)
    assert path.tolist() == [0, 1, 4, 7, 10] and bonus == 999
    # sibling (second child of root) matches, its child does not
    sampled = torch.full((13,), 7)
    sampled[0] = 102
    path, bonus = ops.accept_scan(sampled, spec, parents, mask, want)
    assert path.tolist() == [0, 2] and bonus == 7
    # a deeper node "accepted" but its parent rejected must not be on the path
    sampled = torch.full((13,), 7)
    sampled[1] = 104
    path, _ = ops.accept_scan(sampled, spec, parents, mask, want)
    assert path.tolist() == [0]
    assert ops.first_eos([5, 9, 3], [3, 9]) == 1 and ops.first_eos([5, 6], [3]) == -1


def test_awq_pack_roundtrip_and_linear():
    rs = np.random.RandomState(1)
    q = rs.randint(0, 16, size=(256, 64)).astype(np.uint8)
    assert np.array_equal(ops.awq_unpack(ops.awq_pack(q)), q)
    # nibble i of word c <-> column 8c + ORDER[i]
    one = np.zeros((1, 8), dtype=np.uint8); one[0, 2] = 0xF
    assert ops.awq_pack(one).view(np.uint32)[0, 0] == 0xF << 4     # ORDER[1] == 2
    z = rs.randint(0, 16, size=(2, 64)).astype(np.uint8)
    s = (rs.rand(2, 64) * 0.02 + 0.005).astype(np.float16)
    W = (q.astype(np.float32) - np.repeat(z, 128, 0)) * np.repeat(s.astype(np.float32), 128, 0)
    x = torch.from_numpy(rs.randn(3, 256).astype(np.float32))
    out = ops.awq_linear(x, T(ops.awq_pack(q)), T(ops.awq_pack(z)), T(s), 128)
    torch.testing.assert_close(out, x @ torch.from_numpy(W.astype(np.float16).astype(np.float32)), rtol=1e-4, atol=1e-4)
    from umbrella_amd.models.awq_format import pack_rows, unpack_rows
    assert np.array_equal(pack_rows(q), ops.awq_pack(q)) and np.array_equal(unpack_rows(pack_rows(q)), q)


# ------------------------------------------------------------------ static engine, stochastic verification
def _stochastic_cases():
    with open(os.path.join(GOLD, "engines_stochastic.json")) as f:
        return json.load(f)["cases"]


@pytest.mark.parametrize("case_name", ["static_3x4_stochastic", "static_3x4_selfdraft_stochastic"])
def test_static_stochastic_replays_reference_trace(case_name):
    """The reference's static engine on its sampling path (static:131,298-310; tests/golden/make_golden_stochastic.py):
    with the recorded uniform_samples the oracle engine reproduces every iteration's tree, sampled ids, accept result
    and bonus token -- the penalty -> /temperature -> one-sampler-call-with-the-same-uniforms order is pinned (the
    sampler's own internals are a restatement: flashinfer wheel absent)."""
    from oracle.engine import OracleStaticEngine
    case = _stochastic_cases()[case_name]
    c = case["config"]
    self_draft = "selfdraft" in case_name
    dcfg, dseed = (G["target_cfg"], G["seeds"]["target"]) if self_draft else (G["draft_cfg"], G["seeds"]["draft"])
    L = c["max_length"]
    target = oracle_model(G["target_cfg"], G["seeds"]["target"], L, torch.float32)
    draft = oracle_model(dcfg, dseed, L, torch.float32, slot_cache=True)
    with open(os.path.join(GOLD, "growmaps.json")) as f:
        gm = json.load(f)["3x4"]
    eng = OracleStaticEngine(draft, target, gm, case["eos"], max_length=L, safe_buffer=c["safe_buffer"],
                             temperature=c["temperature"], topp=c["topp"], topk=c["topk"],
                             repetition_penalty=c["repetition_penalty"],
                             uniform_samples=torch.tensor(case["uniform_samples"]))
    assert eng._prefill(torch.tensor([case["prompt"]]))
    assert int(eng.tokens[0, eng.num_nodes]) == case["first_token"]
    start = eng.num_nodes
    for rec in case["iters"]:
        assert eng.num_nodes == rec["n"]
        eng.build_tree()
        assert eng.tokens[0, rec["n"]:rec["n"] + eng.tree_size].tolist() == rec["tree_tokens"]
        go = eng.verify()
        assert eng.trace[-1]["sampled"] == rec["sampled"]
        assert eng.num_nodes == rec["num_nodes"] and go == rec["go_on"]
        assert int(eng.tokens[0, eng.num_nodes]) == rec["bonus"]
    assert eng.tokens[0, start:eng.num_nodes + 1].tolist() == case["tokens"]


def test_rejection_sampler_limit_is_the_renormalised_nucleus():
    """oracle.ops.top_k_top_p_sampling_from_logits (flashinfer's rejection sampler, restated): every accepted draw lies
    in the nucleus, empirical frequencies follow nucleus_distribution (chi-square), and that set is the one
    top_p_renorm keeps -- the distribution umb_sample_rows is tested against on the GPU."""
    g = torch.Generator().manual_seed(3)
    logits = torch.randn(1, 64, generator=g) * 2.0
    k, p = 16, 0.9
    nuc = ops.nucleus_distribution(logits, k, p)[0]
    ref = ops.top_p_renorm(torch.softmax(ops.keep_topk(logits, k), dim=-1), p)[0]
    torch.testing.assert_close(nuc, ref, rtol=1e-5, atol=1e-7)
    n = 4000
    u = torch.rand(32, n, generator=g)                          # 32 rounds: rejection never runs out
    ids, ok = ops.top_k_top_p_sampling_from_logits(logits.expand(n, -1), u, k, p)
    assert bool(ok.all()) and bool((nuc[ids] > 0).all())
    cnt = torch.bincount(ids, minlength=64).float()
    sup = nuc > 0
    chi2 = float((((cnt - n * nuc) ** 2)[sup] / (n * nuc[sup])).sum())
    assert chi2 < 3.0 * int(sup.sum()) + 20, chi2
    # the reference's 3 rounds: draws that exhaust them may fall outside the nucleus, at most (1 - top_p)^3 of them
    ids3, ok3 = ops.top_k_top_p_sampling_from_logits(logits.expand(n, -1), u[:3], k, p)
    assert float((~ok3).float().mean()) <= (1 - p) ** 3 + 0.01
