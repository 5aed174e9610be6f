"""Dynamic (SpecExec-style beam-grown tree) speculation engine on the HIP path.

Same constructor / methods as the reference class
(umbrella/speculation/dynamic_speculation_engine.py:18-544).  The beam expansion
(top-num_beams per node, local log-softmax score accumulation, global top-width,
parent + mask-row propagation, dynamic:236-248) runs in two kernels per level and
never leaves the GPU; the target is layer-offloaded when ``offload`` is true
(the reference hard-codes it, dynamic:77-80; its configs all say "offload": true).
"""
from __future__ import annotations

import torch

from .. import _lib
from .engine_common import HipEngine, logger
from ..utils import TextColors


class DynamicSpeculationEngine(HipEngine):
    MASK_FIRST_EOS = True                       # dynamic:130,163

    def __init__(self, draft_model_name: str, target_model_name: str, dtype=torch.float16, device: str = "cuda:0",
                 **kwargs) -> None:
        super().__init__()
        self.draft_model_name, self.target_model_name = draft_model_name, target_model_name
        self.dtype, self.device = dtype, device
        self.num_beams = kwargs.pop("num_beams", 24)
        self.tree_width = kwargs.pop("width", 16)
        self.tree_depth = kwargs.pop("depth", 24)
        self.offload_target = kwargs.pop("offload", True)
        self._common_kwargs(kwargs)
        self.config = kwargs

    def initialize(self):
        dev, W, Dp = self.device, self.tree_width, self.tree_depth
        T = W * Dp + 1
        assert W * self.num_beams <= 1024 and W <= 64, "beam_expand handles width*num_beams <= 1024"
        logger.info(TextColors.colorize("Tree Size {} | Tree Depth {} | Tree Width {}".format(T - 1, Dp, W), "magenta"))
        self.depth = torch.tensor([0] + [i + 1 for i in range(Dp) for _ in range(W)], dtype=torch.int32, device=dev)
        self.parents = torch.zeros(T, dtype=torch.int32, device=dev)
        self.tree_score = torch.zeros(T, dtype=torch.float32, device=dev)
        self.mask_words = (T + 63) // 64
        self.mask_bits = torch.zeros(T, self.mask_words, dtype=torch.int64, device=dev)
        self.mask_bits[0, 0] = 1                                   # root attends itself
        self.top_idx = torch.zeros(W * self.num_beams, dtype=torch.int32, device=dev)
        self.top_val = torch.zeros(W * self.num_beams, dtype=torch.float32, device=dev)
        # row counters (zeroed once, self-resetting) + per-part keys of the split top-k (umb_topk_rows_ws)
        self.topk_ws = torch.zeros(4096 + W * 16 * self.num_beams * 8, dtype=torch.uint8, device=dev)
        self.draft_rows = W
        self._load_models(dict(offload=False), dict(offload=bool(self.offload_target)))
        if getattr(self.target_model, "_off", None) is not None:
            # the event-ordered copy stream of the streamed verify is launched eagerly; the draft tree (depth + 1 small
            # forwards, top-k and beam kernels: ~1.5 k launches) still replays as one hipGraph
            self.graph_scope = "draft"
        self._alloc_state(T, Dp + 1)
        self.enable_override = False

    @torch.inference_mode()
    def build_tree(self):
        d, W, B = self.draft_model, self.tree_width, self.num_beams
        for step in range(self.tree_depth + 1):
            w = W if step > 0 else 1
            off = 0 if step == 0 else 1 + (step - 1) * W
            last = step == self.tree_depth                          # last forward only fills the draft KV (dynamic:218,235)
            if last and self.lookback:
                break                                               # re-derived by the next root forward (_draft_root)
            if step == 0:
                self._draft_root()
            else:
                d.forward_tree(self.tokens, self.n_dev, self.depth, off, w, self.mask_bits, self.mask_words,
                               head_from=w if last else 0)
            if last:
                break
            _lib.call("umb_topk_rows_ws", self.top_idx, self.top_val, d.logits_buffer, w, self.vocab_size, B,
                      None, None, None, None, self.topk_ws, self.topk_ws.numel())
            _lib.call("umb_beam_expand", self.top_idx, self.top_val, w, B, W, off, self.tree_score, self.parents,
                      self.tokens, self.n_dev, self.mask_bits, self.mask_words)

    def _verify_forward(self):
        self.target_model.forward_tree(self.tokens, self.n_dev, self.depth, 0, self.tree_size, self.mask_bits,
                                       self.mask_words, head_from=0)
