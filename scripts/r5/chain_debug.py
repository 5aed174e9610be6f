"""Bring-up probe of the persistent chain (csrc/chain.hip) through the C ABI, one mode per process:
  python scripts/r5/chain_debug.py tail|front|full [T]
Random 1B-shaped weights; compares against the GEMV launches (umb_gemv) on the same buffers and prints max |diff|."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from umbrella_amd import _lib  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "tail"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
dt = torch.bfloat16 if os.environ.get("DT") == "bf16" else torch.float16
DTC = 1 if dt == torch.bfloat16 else 0
H, I, NQ, Hq, Hkv, D, Lmax = 2048, 8192, 3072, 32, 8, 64, 256
g = torch.Generator(device="cpu").manual_seed(0)


def rnd(*shape, scale=0.02):
    return (torch.randn(*shape, generator=g) * scale).to(dt).to(dev)


lib = _lib.load()
w_o, w_gu, w_dn, w_q = rnd(H, H), rnd(2 * I, H), rnd(H, I), rnd(NQ, H)
if os.environ.get("DN_SLICE"):
    j = int(os.environ["DN_SLICE"])
    keep = w_dn[:, 2048 * j:2048 * (j + 1)].clone()
    w_dn.zero_()
    w_dn[:, 2048 * j:2048 * (j + 1)] = keep
norm2, nnorm = (1 + rnd(H, scale=0.1)), (1 + rnd(H, scale=0.1))
attn, h, hw = rnd(4, H, scale=1.0), rnd(4, H, scale=1.0), rnd(4, H, scale=1.0)
ssq = torch.zeros(4, 256, dtype=torch.float32, device=dev)
ssq[:, :32] = (hw.float() ** 2).view(4, 32, 64).sum(-1)
pos = torch.arange(40, 44, dtype=torch.int32, device=dev)
cosT, sinT = rnd(Lmax, D, scale=1.0), rnd(Lmax, D, scale=1.0)
if os.environ.get("ROPE_TRIVIAL"):
    cosT.fill_(1.0); sinT.fill_(0.0)
if os.environ.get("SSQ_CONST"):
    ssq[:, :32] = 64.0
if os.environ.get("SAME_ROWS"):
    hw[1:] = hw[0]; h[1:] = h[0]; attn[1:] = attn[0]; ssq[1:] = ssq[0]; pos[:] = 40
if os.environ.get("ROLL_ROWS"):
    hw = hw.roll(-1, 0).contiguous(); h = h.roll(-1, 0).contiguous(); attn = attn.roll(-1, 0).contiguous()
    ssq = ssq.roll(-1, 0).contiguous(); pos = pos.roll(-1, 0).contiguous()
EPS = 0.0 if os.environ.get("EPS0") else 1e-5


def bufs():
    return dict(q=torch.zeros(4, Hq * D, dtype=dt, device=dev), kc=torch.zeros(Hkv, Lmax, D, dtype=dt, device=dev),
                vt=torch.zeros(Hkv, D, Lmax + 32, dtype=dt, device=dev), h=h.clone(), hw=hw.clone(), ssq=ssq.clone(),
                act=torch.zeros(4, I, dtype=dt, device=dev))


n = lib.umb_chain_xchg_bytes(4, H, I)
xchg = torch.zeros(n, dtype=torch.uint8, device=dev)
_lib.check(lib.umb_chain_xchg_init(xchg.data_ptr(), 4, H, I, _lib.stream_ptr()))
print("ok", lib.umb_chain_ok(T, H, I, NQ, D, 0), "xchg bytes", n, flush=True)


def chain(b, front, tail):
    c = _lib.UmbChain()
    c.w_o, c.w_gu, c.w_down, c.w_qkv = w_o.data_ptr(), w_gu.data_ptr(), w_dn.data_ptr(), w_q.data_ptr()
    c.attn, c.h, c.hw, c.ssq = attn.data_ptr(), b["h"].data_ptr(), b["hw"].data_ptr(), b["ssq"].data_ptr()
    c.norm2, c.next_norm = norm2.data_ptr(), nnorm.data_ptr()
    c.pos, c.slot, c.cosT, c.sinT = pos.data_ptr(), pos.data_ptr(), cosT.data_ptr(), sinT.data_ptr()
    c.q_out, c.k_cache, c.vt_cache, c.xchg = b["q"].data_ptr(), b["kc"].data_ptr(), b["vt"].data_ptr(), xchg.data_ptr()
    c.T, c.Tmax, c.front, c.tail, c.H, c.I, c.NQKV = T, 4, front, tail, H, I, NQ
    c.ssq_stride, c.ssq_groups_in, c.Hq, c.Hkv, c.D, c.Lmax, c.eps = 256, 32, Hq, Hkv, D, Lmax, EPS
    rc = lib.umb_draft_chain(C.byref(c), DTC, _lib.stream_ptr())
    print("launch rc", rc, flush=True)
    torch.cuda.synchronize()
    st = C.c_uint32(0)
    lib.umb_chain_status(xchg.data_ptr(), 4, H, I, C.byref(st), _lib.stream_ptr())
    print("synchronised, status", hex(st.value), flush=True)


def gemv(b, front, tail):
    def fx(**kw):
        f = _lib.UmbGemmLL()
        for k, v in kw.items():
            setattr(f, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
        return f
    groups = 32
    if front:
        f = fx(h=b["h"], hw=b["hw"], norm_w=norm2, ssq_out=b["ssq"], ssq_out_stride=256)
        _lib.check(lib.umb_gemv(None, attn.data_ptr(), w_o.data_ptr(), T, H, H, 4, C.byref(f), DTC, _lib.stream_ptr()))
        f = fx(ssq_in=b["ssq"], ssq_groups=256, ssq_in_stride=256, ssq_dim=float(H), eps=EPS)
        _lib.check(lib.umb_gemv(b["act"].data_ptr(), b["hw"].data_ptr(), w_gu.data_ptr(), T, 2 * I, H, 2, C.byref(f), DTC, _lib.stream_ptr()))
        f = fx(h=b["h"], hw=b["hw"], norm_w=nnorm, ssq_out=b["ssq"], ssq_out_stride=256)
        _lib.check(lib.umb_gemv(None, b["act"].data_ptr(), w_dn.data_ptr(), T, H, I, 4, C.byref(f), DTC, _lib.stream_ptr()))
        groups = 256
    if tail:
        f = fx(ssq_in=b["ssq"], ssq_groups=groups, ssq_in_stride=256, ssq_dim=float(H), eps=EPS, pos=pos, slot=pos, cosT=cosT,
               sinT=sinT, q_out=b["q"], k_cache=b["kc"], vt_cache=b["vt"], Hq=Hq, Hkv=Hkv, D=D, Lmax=Lmax)
        _lib.check(lib.umb_gemv(None, b["hw"].data_ptr(), w_q.data_ptr(), T, NQ, H, 3, C.byref(f), DTC, _lib.stream_ptr()))
    torch.cuda.synchronize()


front, tail = {"tail": (0, 1), "front": (1, 0), "full": (1, 1)}[mode]
a, b = bufs(), bufs()
gemv(b, front, tail)
print("gemv reference done", flush=True)
chain(a, front, tail)
for k in ("h", "hw", "ssq", "q", "kc", "vt"):
    x, y = a[k].float(), b[k].float()
    if k in ("h", "hw", "ssq"):
        x, y = x[:T], y[:T]
    d = (x - y).abs()
    if int((d > 0).sum()) and int((d > 0).sum()) < 20:
        for idx in (d > 0).nonzero().tolist():
            print("   mismatch at", idx, "chain", float(x[tuple(idx)]), "gemv", float(y[tuple(idx)]), flush=True)
    if k == "h" and int((d > 0).sum()) >= 20:
        nz0 = (d > 0).nonzero()
        same_as_input = sum(float(x[tuple(i)]) == float(h.float()[tuple(i)]) for i in nz0.tolist())
        print(f"   of the {len(nz0)} mismatching h entries, {same_as_input} still hold the INPUT h (never stored)", flush=True)
    if k in ("h", "q") and int((d > 0).sum()) >= 20:
        nz = (d > 0).nonzero()
        import collections
        print("   tokens", dict(collections.Counter(nz[:, 0].tolist())), "col % 8" if k == "h" else "col % 12",
              dict(collections.Counter((nz[:, 1] % (8 if k == "h" else 12)).tolist())), flush=True)
    print(f"{mode} T={T} {k:4s}: max|diff| {float(d.max()):.3e}  mismatches {int((d > 0).sum())} / {d.numel()}  ref max {float(y.abs().max()):.3e}",
          flush=True)

if os.environ.get("SAME_ROWS"):
    for name, bb in (("chain", a), ("gemv", b)):
        qq = bb["q"].float()
        for t in range(1, T):
            d = (qq[t] - qq[0]).abs()
            print(f"{name}: token {t} vs token 0 (identical inputs): mismatches {int((d > 0).sum())}", (d > 0).nonzero().flatten().tolist()[:8], flush=True)
