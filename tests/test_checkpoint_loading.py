"""Checkpoint loading the way the reference does it (umbrella/models/llama.py:229-236 HF from_pretrained,
llama_layer.py:220-258, quantization/awq_utils.py:20-36 AutoAWQ qweight / qzeros / scales): a local Hugging Face
directory (config.json [+ generation_config.json] + *.safetensors, possibly sharded) through
AutoModelLM.from_pretrained(<dir>) must give exactly the model the injected state dict gives."""
import json
import os

import pytest
import torch

from helpers import load_golden

G = load_golden()


def _write_hf_dir(path, cfgd, sd, eos, awq=False, shards=1, model_type="llama"):
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    c = {k: v for k, v in cfgd.items() if k not in ("rope_scaling", "rope_theta")}
    c.update(model_type=model_type, architectures=["LlamaForCausalLM"], max_position_embeddings=131072,
             rope_parameters=dict(cfgd["rope_scaling"], rope_theta=cfgd["rope_theta"]), eos_token_id=eos[0])
    if awq:
        c["quantization_config"] = {"quant_method": "awq", "bits": 4, "group_size": 128, "zero_point": True, "version": "gemm"}
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(c, f)
    with open(os.path.join(path, "generation_config.json"), "w") as f:
        json.dump({"eos_token_id": eos}, f)
    names = sorted(sd)
    per = (len(names) + shards - 1) // shards
    for i in range(shards):
        part = {n: sd[n].contiguous() for n in names[i * per:(i + 1) * per]}
        fn = "model.safetensors" if shards == 1 else f"model-{i + 1:05d}-of-{shards:05d}.safetensors"
        save_file(part, os.path.join(path, fn))


def _state(cfgd, seed, awq, dtype):
    from umbrella_amd.models.config import LlamaCfg
    from umbrella_amd.models.synthetic import synth_awq_small, synth_state_small
    cfg = LlamaCfg(**dict(cfgd, awq=awq))
    sd = synth_awq_small(cfg, seed) if awq else synth_state_small(cfg, seed)
    # a real checkpoint stores 16-bit dense tensors; AWQ triples are int32 / int32 / fp16
    return {k: (v.to(dtype) if v.dtype == torch.float32 else v) for k, v in sd.items()}


def test_config_from_hf_directory(tmp_path):
    """CPU: config.json + generation_config.json of a local directory -> LlamaCfg (dims, llama3 rope scaling from
    transformers>=5 `rope_parameters`, eos from GenerationConfig as the engines read it (static:104-108), AWQ flag)."""
    from umbrella_amd.models.config import LlamaCfg
    d = str(tmp_path / "tiny-awq")
    sd = _state(G["target_cfg"], 11, True, torch.float16)
    _write_hf_dir(d, G["target_cfg"], sd, [3, 5], awq=True, shards=2)
    cfg = LlamaCfg.from_dir(d)
    t = G["target_cfg"]
    assert (cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers, cfg.vocab_size) == \
        (t["hidden_size"], t["intermediate_size"], t["num_hidden_layers"], t["vocab_size"])
    assert cfg.head_dim == t["head_dim"] and cfg.num_key_value_heads == t["num_key_value_heads"]
    assert cfg.awq and cfg.awq_group == 128 and cfg.eos_token_id == [3, 5]
    assert cfg.rope_theta == t["rope_theta"] and cfg.rope_scaling["rope_type"] == "llama3"
    assert cfg.tie_word_embeddings is False
    assert sorted(f for f in os.listdir(d) if f.endswith(".safetensors")) == \
        ["model-00001-of-00002.safetensors", "model-00002-of-00002.safetensors"]


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["dense", "dense_tied_sharded", "awq"])
def test_from_pretrained_directory_equals_state_dict(tmp_path, kind):
    """AutoModelLM.from_pretrained(<dir>) (safetensors on disk, dense HF names or AutoAWQ qweight / qzeros / scales)
    == the same tensors injected as a state dict: packed weights byte for byte, logits bit for bit."""
    from umbrella_amd.models import AutoModelLM
    from umbrella_amd.models.config import LlamaCfg
    from umbrella_amd.models.llama import Llama
    dev, dtype = "cuda:0", torch.float16
    awq = kind == "awq"
    cfgd, seed = (G["draft_cfg"], 22) if kind == "dense_tied_sharded" else (G["target_cfg"], 11)
    sd = _state(cfgd, seed, awq, dtype)
    d = str(tmp_path / kind)
    _write_hf_dir(d, cfgd, sd, [3, 5], awq=awq, shards=3 if "sharded" in kind else 1)
    if cfgd["tie_word_embeddings"]:
        assert "lm_head.weight" not in sd                        # tied: the head is read from the embedding table

    m1 = AutoModelLM.from_pretrained(d, max_length=128, device=dev, dtype=dtype)
    assert isinstance(m1, Llama) and m1.config.awq == awq and m1.eos_tokens == [3, 5]
    m1.alloc()
    m2 = Llama("tiny", max_length=128, device=dev, dtype=dtype, state_dict=sd,
               config=LlamaCfg(**dict(cfgd, eos_token_id=[3, 5], awq=awq)))
    m2.alloc()
    for l1, l2 in zip(m1.layers, m2.layers):
        for key in ("qkv", "o", "gu", "down"):
            assert torch.equal(l1[key].w, l2[key].w)
            if awq:
                assert torch.equal(l1[key].meta, l2[key].meta)
    assert torch.equal(m1.embed_tokens, m2.embed_tokens) and torch.equal(m1.lm_head.w, m2.lm_head.w)
    T = 9
    ids = torch.randint(6, cfgd["vocab_size"], (1, T), generator=torch.Generator().manual_seed(1))
    pos = torch.arange(T)[None]
    mask = torch.tril(torch.ones(T, 128, dtype=torch.bool))
    a = m1.inference(ids, pos, mask, torch.arange(T))
    b = m2.inference(ids, pos, mask, torch.arange(T))
    assert torch.equal(a, b) and bool(torch.isfinite(a).all())
    # cuda_graph / offload flags route to the same loader (auto_model.py:165-182)
    m3 = AutoModelLM.from_pretrained(d, offload=True, max_length=128, device=dev, dtype=dtype)
    m3.alloc(num_cache_layers=1)
    assert m3._off is not None
    c = m3.inference(ids, pos, mask, torch.arange(T))
    assert torch.equal(a, c)


def test_unknown_name_raises(tmp_path):
    from umbrella_amd.models import AutoModelLM
    with pytest.raises(ValueError):
        AutoModelLM.from_pretrained("nobody/No-Such-Model")
    with pytest.raises(ValueError):
        AutoModelLM.from_pretrained(str(tmp_path))               # a directory without config.json is not a checkpoint
