"""Dropout on the HIP training path (train_vit.py:100,107,125; torch functional.py:6569-6570).

torch's Philox masks cannot be reproduced by another kernel, so parity is checked
by REPLAY: the masks the kernels drew are read back through rvit_dropout_mask with the step's site
keys, handed to the oracle (oracle/ref_dropout.py: the reference forward with explicit masks), and
loss + every parameter gradient are compared with autograd under the same stated tolerances as the
dropout-free training step (tests/test_train_gpu.py).  Statistical properties of the generator
(keep rate, independence across sites, determinism) are checked separately.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def test_mask_statistics(dev):
    from vit_amd import ops
    n, p = 1 << 22, 0.1
    a = ops.dropout_mask(n, p, 0x1234_5678_9ABC_DEF0, dev).float()
    b = ops.dropout_mask(n, p, 0x0FED_CBA9_8765_4321, dev).float()
    a2 = ops.dropout_mask(n, p, 0x1234_5678_9ABC_DEF0, dev).float()
    assert torch.equal(a, a2)                                   # deterministic in (p, seed)
    sigma = (p * (1 - p) / n) ** 0.5
    for m in (a, b):
        assert abs(m.mean().item() - (1 - p)) < 5 * sigma       # keep rate = 1 - p
    # independent sites: P(both kept) = (1-p)^2
    both = (a * b).mean().item()
    assert abs(both - (1 - p) ** 2) < 6 * sigma
    # no short-range structure inside a site: lag-1 and lag-64 agreement at the independent rate
    for lag in (1, 64, 4096):
        j = (a[:-lag] * a[lag:]).mean().item()
        assert abs(j - (1 - p) ** 2) < 6 * sigma, (lag, j)
    assert ops.dropout_mask(1000, 0.0, 5, dev).all()            # p = 0 keeps everything
    for pp in (0.25, 0.5, 0.9):
        m = ops.dropout_mask(n, pp, 77, dev).float().mean().item()
        assert abs(m - (1 - pp)) < 5 * (pp * (1 - pp) / n) ** 0.5


@pytest.mark.parametrize("n,p,seed", [(1 << 16, 0.1, 0x1234_5678_9ABC_DEF0), (12345, 0.3, 7), (2, 0.5, 0xFFFF_FFFF_FFFF_FFFF),
                                      (1 << 20, 0.05, 0x0123_4567_0000_0001), (777, 0.9, 1 << 63)])
def test_mask_equals_the_numpy_restatement(dev, n, p, seed):
    """Integer work, bit-exact bar: the generator of csrc/common.h against oracle/ref_dropout.keep_mask_numpy."""
    import numpy as np
    from oracle.ref_dropout import keep_mask_numpy
    from vit_amd import ops
    got = ops.dropout_mask(n, p, seed, dev).cpu().numpy()
    want = keep_mask_numpy(n, p, seed)
    assert got.shape == want.shape and np.array_equal(got, want), f"{int((got != want).sum())} of {n} decisions differ"


def test_dropout_apply_matches_mask(dev):
    from vit_amd import ops
    rows, cols, p, seed = 300, 256, 0.1, 991
    x = torch.randn(rows, cols, device=dev)
    m = ops.dropout_mask(rows * cols, p, seed, dev).view(rows, cols).float()
    y = ops.dropout_apply(x, torch.empty_like(x), p, seed)
    assert torch.allclose(y, x * m / (1 - p), rtol=1e-6, atol=0)
    xb = x.to(torch.bfloat16)
    yb = ops.dropout_apply(xb, torch.empty_like(xb), p, seed)
    assert torch.allclose(yb.float(), (xb.float() * m / (1 - p)).to(torch.bfloat16).float(), rtol=1e-2, atol=1e-3)
    # pitched view (a column block of a wider matrix) and in-place
    wide = torch.randn(rows, 3 * cols, device=dev).to(torch.bfloat16)
    blk = wide[:, cols:2 * cols]
    want = (blk.float() * m / (1 - p)).to(torch.bfloat16)
    ops.dropout_apply(blk, blk, p, seed)
    assert torch.allclose(blk.float(), want.float(), rtol=1e-2, atol=1e-3)


def test_gemm_epilogue_dropout_positions(dev):
    from vit_amd import ops
    M, N, K, p, seed = 200, 128, 64, 0.25, 4242
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) * 0.2).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev)
    m = ops.dropout_mask(M * N, p, seed, dev).view(M, N).float() / (1 - p)
    base = a.float() @ w.float().t() + bias
    out = torch.empty(M, N, device=dev)
    ops.gemm(a, w, out, bias=bias, residual=res, drop=("pre_residual", p, seed))
    assert torch.allclose(out, base * m + res, rtol=2e-2, atol=2e-2)
    ops.gemm(a, w, out, bias=bias, act="gelu", drop=("post_act", p, seed))
    assert torch.allclose(out, F.gelu(base) * m, rtol=2e-2, atol=2e-2)
    # ragged N (scalar epilogue)
    N2 = 44
    out2 = torch.empty(M, N2, device=dev)
    m2 = ops.dropout_mask(M * N2, p, seed, dev).view(M, N2).float() / (1 - p)
    ops.gemm(a, w[:N2].contiguous(), out2, bias=bias[:N2].contiguous(), act="gelu", drop=("post_act", p, seed))
    assert torch.allclose(out2, F.gelu(base[:, :N2]) * m2, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("S,H", [(50, 2), (197, 3)])
def test_attention_dropout_fwd_bwd(dev, S, H):
    from vit_amd import ops
    B, p, seed = 3, 0.2, 31337
    E = H * 64
    g = torch.Generator().manual_seed(S)
    qkv = (torch.randn(B * S, 3 * E, generator=g) * 0.7).to(dev)
    dout = torch.randn(B * S, E, generator=g).to(dev)
    qb, db = qkv.to(torch.bfloat16), dout.to(torch.bfloat16)
    out = torch.empty(B * S, E, device=dev, dtype=torch.bfloat16)
    lse = torch.empty(B * H * S, device=dev)
    ops.attention(qb, out, B, S, H, lse=lse, drop_p=p, drop_seed=seed)
    dqkv = torch.empty_like(qb)
    ops.attention_bwd(qb, out, db, lse, dqkv, B, S, H, drop_p=p, drop_seed=seed)
    m = ops.attention_dropout_mask(B, H, S, p, seed, dev).float() / (1 - p)
    x = qb.float().requires_grad_(True)
    q, k, v = (u.view(B, S, H, 64).transpose(1, 2) for u in x.chunk(3, dim=-1))
    a = torch.softmax(q @ k.transpose(-1, -2) / 8.0, dim=-1) * m
    ref = (a @ v).transpose(1, 2).reshape(B * S, E)
    ref.backward(db.float())
    assert torch.allclose(out.float(), ref, rtol=3e-2, atol=3e-2)
    rel = ((dqkv.float() - x.grad).norm() / x.grad.norm()).item()
    assert rel < 0.02, rel


CASES = [
    ("vit-small-patch16", dict(num_classes=11, embed_dim=128, num_heads=2, num_layers=2, dropout=0.1, backbone="patch16"),
     6, 0.05, 0.995),
    ("R-hybrid", dict(num_classes=11, embed_dim=256, num_heads=4, num_layers=4, dropout=0.1, backbone="resnet18"),
     8, 0.06, 0.90),
]


def _replay_masks(m, ref, B, dev):
    from oracle.ref_dropout import mask_shapes
    from vit_amd import ops
    from vit_amd.trainer import (SITE_ATTN, SITE_FC2, SITE_FFN, SITE_HEAD, SITE_POS, SITE_PROJ, _layer_site,
                                 _site_seed)
    p, base = m.train_engine._bufs[B]["drop"]
    which = {"attn": SITE_ATTN, "proj": SITE_PROJ, "ffn": SITE_FFN, "fc2": SITE_FC2}
    masks = {}
    for site, shape in mask_shapes(ref, B).items():
        sid = {"pos": SITE_POS, "head": SITE_HEAD}[site] if isinstance(site, str) else _layer_site(site[0], which[site[1]])
        if not isinstance(site, str) and site[1] == "attn":    # [B, H, S, S]: rows padded to an even stride
            masks[site] = ops.attention_dropout_mask(shape[0], shape[1], shape[2], p, _site_seed(base, sid), dev).float().cpu()
            continue
        n = 1
        for d in shape:
            n *= d
        masks[site] = ops.dropout_mask(n, p, _site_seed(base, sid), dev).view(shape).float().cpu()
    return p, masks


@pytest.mark.parametrize("name,cfg,B,grad_rel,grad_cos", CASES, ids=[c[0] for c in CASES])
def test_training_step_with_dropout_matches_mask_replay(dev, name, cfg, B, grad_rel, grad_cos):
    from oracle.ref_dropout import forward_with_masks
    from oracle.ref_model import CNNViTHybridRef
    from vit_amd.loss import cross_entropy
    from vit_amd.model import CNNViTHybrid
    from vit_amd.synth import micro_doppler_batch, synth_state_dict
    torch.set_num_threads(16)
    torch.manual_seed(1234)
    m = CNNViTHybrid(**cfg)
    sd = synth_state_dict({k: v.shape for k, v in m.state_dict().items()}, 4)
    m.load_state_dict(sd)
    m = m.to(dev).train()
    ref = CNNViTHybridRef(**cfg)
    ref.load_state_dict(sd)
    ref.train()
    x = micro_doppler_batch(B, seed=21)
    y = torch.randint(0, 11, (B,), generator=torch.Generator().manual_seed(3))
    logits = m(x.to(dev))
    loss = cross_entropy(logits, y.to(dev), label_smoothing=0.05)
    loss.backward()
    p, masks = _replay_masks(m, ref, B, dev)
    assert p == cfg["dropout"]
    kept = sum(float(v.sum()) for v in masks.values()) / sum(v.numel() for v in masks.values())
    assert abs(kept - (1 - p)) < 0.01
    loss_ref = F.cross_entropy(forward_with_masks(ref, x, masks, p), y, label_smoothing=0.05)
    loss_ref.backward()
    # the same step without dropout must differ (the masks are really applied)
    with torch.no_grad():
        ones = {k: torch.ones_like(v) for k, v in masks.items()}
        loss_nodrop = F.cross_entropy(forward_with_masks(ref, x, ones, 0.0), y, label_smoothing=0.05)
    print(f"\n[{name}] loss hip {loss.item():.5f} replay {loss_ref.item():.5f} (no dropout {loss_nodrop.item():.5f})")
    assert abs(loss.item() - loss_ref.item()) <= 0.02 * max(1.0, abs(loss_ref.item()))
    e_bf16 = {}
    if cfg["backbone"] != "patch16":
        from oracle.ref_bf16_train import attach_bf16_rounding
        ref2 = CNNViTHybridRef(**cfg)
        ref2.load_state_dict(sd)
        ref2.train()
        attach_bf16_rounding(ref2)
        F.cross_entropy(forward_with_masks(ref2, x, masks, p), y, label_smoothing=0.05).backward()
        g0 = dict(ref.named_parameters())
        for n, q in ref2.named_parameters():
            e_bf16[n] = ((q.grad - g0[n].grad).norm() / g0[n].grad.norm().clamp_min(1e-12)).item()
    gref = dict(ref.named_parameters())
    worst = []
    for n, q in m.named_parameters():
        g, r = q.grad.float().cpu(), gref[n].grad
        rel = ((g - r).norm() / r.norm().clamp_min(1e-12)).item()
        cos = F.cosine_similarity(g.flatten(), r.flatten(), dim=0).item()
        worst.append((rel, cos, n))
    worst.sort(reverse=True)
    for rel, cos, n in worst[:5]:
        print(f"   {n:55s} rel {rel:.4f} cos {cos:.5f}  (oracle bf16 sensitivity {e_bf16.get(n, 0.0):.4f})")
    bad = [(n, rel, cos) for rel, cos, n in worst if rel > max(grad_rel, 1.6 * e_bf16.get(n, 0.0)) or cos < grad_cos]
    assert not bad, f"{name}: gradients out of tolerance: {bad[:5]}"


def test_dropout_is_off_in_eval_and_varies_per_step(dev):
    from vit_amd.model import CNNViTHybrid
    from vit_amd.synth import micro_doppler_batch, synth_state_dict
    cfg = CASES[0][1]
    m = CNNViTHybrid(**cfg)
    m.load_state_dict(synth_state_dict({k: v.shape for k, v in m.state_dict().items()}, 2))
    m = m.to(dev)
    x = micro_doppler_batch(4, seed=5).to(dev)
    m.eval()
    with torch.no_grad():
        e1, e2 = m(x), m(x)
    assert torch.equal(e1, e2)
    m.train()
    with torch.no_grad():
        torch.manual_seed(7)
        t1 = m(x).clone()
        t2 = m(x).clone()
        torch.manual_seed(7)
        t3 = m(x).clone()
    assert not torch.equal(t1, t2)          # a fresh mask every step
    assert torch.equal(t1, t3)              # reproducible from torch.manual_seed
    assert not torch.allclose(t1, e1)
