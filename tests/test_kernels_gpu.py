"""GPU parity of each librvit kernel against a plain torch fp32 CPU computation of the same op
on the same (bf16-rounded) inputs.  Tolerances are bf16-output tolerances: the kernels accumulate
in fp32, so the only error sources are the final bf16 rounding (rel 2^-8) and summation order."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _bf(t):
    return t.to(torch.bfloat16)


def _close(got, ref, rtol, atol, what=""):
    got = got.float().cpu()
    ref = ref.float()
    err = (got - ref).abs()
    bound = atol + rtol * ref.abs()
    bad = (err > bound).sum().item()
    assert bad == 0, f"{what}: {bad}/{err.numel()} out of tolerance, max err {err.max().item():.4g}"


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 6, 7])
@pytest.mark.parametrize("M,N,K", [(300, 256, 256), (128, 128, 64), (1000, 768, 200), (77, 11, 128), (64, 1024, 264)])
def test_gemm_plain(dev, M, N, K, tile):
    from vit_amd import ops
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a = _bf(torch.randn(M, K, generator=g))
    w = _bf(torch.randn(N, K, generator=g) / math.sqrt(K))
    ref = a.float() @ w.float().T
    out = torch.empty(M, N, dtype=torch.float32, device=dev)
    ops.gemm(a.to(dev), w.to(dev), out, tile=tile)
    _close(out, ref, 1e-4, 1e-4, "gemm f32 out")
    outb = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ops.gemm(a.to(dev), w.to(dev), outb, tile=tile)
    _close(outb, ref, 1.0 / 128, 1e-3, "gemm bf16 out")


@pytest.mark.parametrize("M,N,K,auto", [(700, 520, 328, False), (256, 256, 64, False), (1000, 72, 1024, False),
                                        (8192 + 40, 8192, 512, True), (50432, 768, 2048, True)])
def test_gemm_256x256_tile_bit_identical(dev, M, N, K, auto):
    """The 256x256 tile (tile 4; what large encoder Linears run) contracts K in the same order as the 256x64 tile:
    identical bits through every epilogue form, ragged M / N / K tiles included; `auto` shapes are ones the
    dispatcher sends to it by itself."""
    from vit_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    a = _bf(torch.randn(M, K, generator=g)).to(dev)
    w = _bf(torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev)
    bias, scale = torch.randn(N, generator=g).to(dev), (torch.rand(N, generator=g) + 0.5).to(dev)
    res = torch.randn(M, N, generator=g).to(dev)
    outs = []
    for tile in (1, 0 if auto else 4):
        o1 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        pre = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        ops.gemm(a, w, o1, bias=bias, scale=scale, residual=res, act="gelu", preact=pre, tile=tile)
        o2 = torch.empty(M, N, dtype=torch.float32, device=dev)
        ops.gemm(a, w, o2, tile=tile)
        outs.append((o1, pre, o2))
    for x, y in zip(*outs):
        assert torch.equal(x, y)
    if M <= 1024:
        _close(outs[1][2], a.float().cpu() @ w.float().cpu().T, 1e-4, 1e-4, "256x256 f32 out")


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (700, 520, 384), (1000, 72, 1024), (513, 264, 256), (77, 24, 128),
                                   (4096, 1024, 768), (20000, 1288, 256), (50432, 768, 3072)])
def test_gemm_8phase_tile_bit_identical(dev, M, N, K):
    """The eight-wave 256x256 kernel with the four-phase K tile (tile 5, csrc/gemm8p.hip; what the ViT-B / ViT-L Linears
    run) against the 256x64 tile: same K order, so identical bits through every epilogue form; ragged M / N tiles, one
    to many output tiles per persistent workgroup (the staging stream runs on into the next output tile), K loops of 2
    to 48 K tiles, and repeated launches (a race between the DMA ring and the fragment reads would show as rare
    differing tiles)."""
    from vit_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    a = _bf(torch.randn(M, K, generator=g)).to(dev)
    w = _bf(torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev)
    bias, scale = torch.randn(N, generator=g).to(dev), (torch.rand(N, generator=g) + 0.5).to(dev)
    res = torch.randn(M, N, generator=g).to(dev)
    outs = []
    for tile in (1, 5):
        o1 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        pre = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        ops.gemm(a, w, o1, bias=bias, scale=scale, residual=res, act="gelu", preact=pre, tile=tile)
        o2 = torch.empty(M, N, dtype=torch.float32, device=dev)
        ops.gemm(a, w, o2, tile=tile)
        outs.append((o1, pre, o2))
    for x, y in zip(*outs):
        assert torch.equal(x, y)
    for _ in range(5 if M > 4096 else 20):
        o3 = torch.empty(M, N, dtype=torch.float32, device=dev)
        ops.gemm(a, w, o3, tile=5)
        assert torch.equal(o3, outs[0][2])
    if M <= 1024:
        _close(outs[1][2], a.float().cpu() @ w.float().cpu().T, 1e-4, 1e-4, "8-phase f32 out")


@pytest.mark.parametrize("form", ["gate_gelu_grad", "gate_relu_bf16res", "drop_pre", "drop_post", "periodic_res_remap",
                                  "lean_plain", "lean_bias", "lean_bias_gelu", "lean_bias_gelu_preact", "lean_bias_res",
                                  "scale_only"])
def test_gemm_8phase_epilogue_forms(dev, form):
    """Every epilogue form through the split (load / apply) epilogue of the eight-wave kernel equals the 256x64 tile's
    single-function epilogue bit for bit: gates of a saved pre-activation, bf16 and periodic residuals, both dropout
    positions, the output row remap; ragged M and N."""
    from vit_amd import ops
    M, N, K = 4500, 1288, 256      # 18 x 6 tiles of 256 x 256: no workgroup runs out of tiles after one
    g = torch.Generator().manual_seed(len(form))
    a = _bf(torch.randn(M, K, generator=g)).to(dev)
    w = _bf(torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    kw = dict(bias=bias)
    rows_out = M
    if form.startswith("lean") or form == "scale_only":   # the compile-time epilogue forms of csrc/gemm8p.hip
        kw = {} if form == "lean_plain" else dict(bias=bias)
        if "gelu" in form:
            kw["act"] = "gelu"
        if "preact" in form:
            kw["preact"] = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        if "res" in form:
            kw["residual"] = torch.randn(M, N, generator=g).to(dev)
        if form == "scale_only":
            kw = dict(scale=(torch.rand(N, generator=g) + 0.5).to(dev))
    elif form == "gate_gelu_grad":
        kw.update(gate=_bf(torch.randn(M, N, generator=g)).to(dev), gate_mode="gelu_grad")
    elif form == "gate_relu_bf16res":
        kw.update(gate=_bf(torch.randn(M, N, generator=g)).to(dev), gate_mode="relu_mask",
                  residual=_bf(torch.randn(M, N, generator=g)).to(dev), act="relu")
    elif form == "drop_pre":
        kw.update(residual=torch.randn(M, N, generator=g).to(dev), drop=("pre_residual", 0.1, 0x1234567887654321))
    elif form == "drop_post":
        kw.update(act="gelu", preact=torch.empty(M, N, dtype=torch.bfloat16, device=dev), drop=("post_act", 0.25, 77))
    else:
        kw.update(residual=torch.randn(50, N, generator=g).to(dev), residual_period=50, row_group=50, row_extra=1)
        rows_out = (M // 50) * 51
    outs = []
    for tile in (1, 5):
        res = []
        for dt in (torch.float32, torch.bfloat16):
            o = torch.zeros(rows_out, N, dtype=dt, device=dev)
            if "preact" in kw:
                kw["preact"] = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            ops.gemm(a, w, o, tile=tile, **kw)
            res.append(o)
            if "preact" in kw:
                res.append(kw["preact"])
        outs.append(res)
    for x, y in zip(*outs):
        assert torch.equal(x, y), form


def test_gemm_identity_asymmetric(dev):
    """A = I with an asymmetric B catches a transposed C write (guide §3)."""
    from vit_amd import ops
    n = 128
    a = _bf(torch.eye(n))
    w = _bf(torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 251 - 100)
    out = torch.empty(n, n, dtype=torch.float32, device=dev)
    ops.gemm(a.to(dev), w.to(dev), out)
    assert torch.equal(out.cpu(), w.float().T)


@pytest.mark.parametrize("act", [None, "relu", "gelu"])
@pytest.mark.parametrize("res_dtype", [None, torch.float32, torch.bfloat16])
def test_gemm_epilogue(dev, act, res_dtype):
    from vit_amd import ops
    M, N, K = 200, 320, 128
    g = torch.Generator().manual_seed(5)
    a = _bf(torch.randn(M, K, generator=g))
    w = _bf(torch.randn(N, K, generator=g) / math.sqrt(K))
    bias = torch.randn(N, generator=g)
    scale = torch.rand(N, generator=g) + 0.5
    res = None if res_dtype is None else torch.randn(M, N, generator=g).to(res_dtype)
    pre = a.float() @ w.float().T * scale + bias
    ref = pre if res is None else pre + res.float()
    if act == "relu":
        ref = torch.relu(ref)
    elif act == "gelu":
        ref = F.gelu(ref)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    preact = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ops.gemm(a.to(dev), w.to(dev), out, bias=bias.to(dev), scale=scale.to(dev),
             residual=None if res is None else res.to(dev), act=act, preact=preact)
    _close(out, ref, 1.0 / 128, 2e-3, "epilogue out")
    _close(preact, pre, 1.0 / 128, 2e-3, "preact")


def test_gemm_scalar_tail_and_inplace_residual(dev):
    from vit_amd import ops
    M, N, K = 33, 11, 128
    g = torch.Generator().manual_seed(9)
    a = _bf(torch.randn(M, K, generator=g))
    w = _bf(torch.randn(N, K, generator=g) / 8)
    bias = torch.randn(N, generator=g)
    out = torch.empty(M, N, dtype=torch.float32, device=dev)
    ops.gemm(a.to(dev), w.to(dev), out, bias=bias.to(dev))
    _close(out, a.float() @ w.float().T + bias, 1e-4, 1e-4, "N=11")
    # in-place fp32 residual stream update: x = x + a @ w.T + b
    N2 = 256
    w2 = _bf(torch.randn(N2, K, generator=g) / 8)
    b2 = torch.randn(N2, generator=g)
    x = torch.randn(M, N2, generator=g)
    xd = x.to(dev)
    ops.gemm(a.to(dev), w2.to(dev), xd, bias=b2.to(dev), residual=xd)
    _close(xd, x + a.float() @ w2.float().T + b2, 1e-4, 1e-4, "in-place residual")


def test_gemm_patch_embed_remap(dev):
    """1x1 patch_embed epilogue: token-major scatter past the cls row + pos_embed add
    (train_vit.py:149-159)."""
    from vit_amd import ops
    B, P, C, E = 5, 49, 512, 256
    g = torch.Generator().manual_seed(11)
    feat = _bf(torch.randn(B * P, C, generator=g))
    w = _bf(torch.randn(E, C, generator=g) / math.sqrt(C))
    bias = torch.randn(E, generator=g)
    pos = torch.randn(P + 1, E, generator=g)
    x = torch.full((B * (P + 1), E), 7.0, dtype=torch.float32, device=dev)
    posd = pos.to(dev)
    ops.gemm(feat.to(dev), w.to(dev), x, bias=bias.to(dev), residual=posd[1:], residual_period=P,
             row_group=P, row_extra=1)
    ref = (feat.float() @ w.float().T + bias).reshape(B, P, E) + pos[1:]
    got = x.cpu().reshape(B, P + 1, E)
    assert torch.all(got[:, 0] == 7.0), "cls rows must be untouched"
    _close(got[:, 1:], ref, 1e-4, 1e-4, "patch rows")


@pytest.mark.parametrize("rows,E", [(37, 256), (1000, 768), (5, 1024), (12, 384), (9, 64)])
@pytest.mark.parametrize("in_dt", [torch.float32, torch.bfloat16])
def test_layernorm(dev, rows, E, in_dt):
    from vit_amd import ops
    g = torch.Generator().manual_seed(rows + E)
    x = (torch.randn(rows, E, generator=g) * 2 + 0.5).to(in_dt)
    gam = torch.randn(E, generator=g)
    bet = torch.randn(E, generator=g)
    ref = F.layer_norm(x.float(), (E,), gam, bet, 1e-5)
    out = torch.empty(rows, E, dtype=torch.float32, device=dev)
    mean = torch.empty(rows, dtype=torch.float32, device=dev)
    rstd = torch.empty(rows, dtype=torch.float32, device=dev)
    ops.layernorm(x.to(dev), gam.to(dev), bet.to(dev), out, mean=mean, rstd=rstd)
    _close(out, ref, 1e-5, 2e-5, "ln f32")
    _close(mean, x.float().mean(1), 1e-5, 1e-5, "mean")
    _close(rstd, 1 / torch.sqrt(x.float().var(1, unbiased=False) + 1e-5), 1e-5, 1e-5, "rstd")
    outb = torch.empty(rows, E, dtype=torch.bfloat16, device=dev)
    ops.layernorm(x.to(dev), gam.to(dev), bet.to(dev), outb)
    _close(outb, ref, 1.0 / 128, 1e-3, "ln bf16")


def test_layernorm_strided_cls_rows(dev):
    from vit_amd import ops
    B, S, E = 6, 50, 256
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, S, E, generator=g)
    gam, bet = torch.randn(E, generator=g), torch.randn(E, generator=g)
    xd = x.to(dev)
    out = torch.empty(B, E, dtype=torch.float32, device=dev)
    ops.layernorm(xd.view(B, S * E)[:, :E], gam.to(dev), bet.to(dev), out)
    _close(out, F.layer_norm(x[:, 0], (E,), gam, bet), 1e-5, 2e-5, "cls LN")


@pytest.mark.parametrize("B,S,H", [(3, 50, 4), (2, 197, 12), (5, 17, 2), (1, 64, 1), (2, 100, 3), (1, 256, 2),
                                   (3, 130, 2), (2, 161, 1)])
def test_attention(dev, B, S, H):
    from vit_amd import ops
    E = H * 64
    g = torch.Generator().manual_seed(B * 100 + S)
    qkv = _bf(torch.randn(B * S, 3 * E, generator=g) * 1.5)
    q, k, v = [t.reshape(B, S, H, 64).transpose(1, 2).float() for t in qkv.split(E, dim=1)]
    sc = q @ k.transpose(-1, -2) / 8.0
    ref = (torch.softmax(sc, -1) @ v).transpose(1, 2).reshape(B * S, E)
    ref_lse = torch.logsumexp(sc, -1)
    out = torch.zeros(B * S, E, dtype=torch.bfloat16, device=dev)
    lse = torch.empty(B, H, S, dtype=torch.float32, device=dev)
    ops.attention(qkv.to(dev), out, B, S, H, lse=lse)
    # P is rounded to bf16 before PV: abs error ~ 2^-9 * max|v| per row, plus the output rounding
    _close(out, ref, 1.0 / 64, 2e-2, "attention out")
    _close(lse, ref_lse, 1e-4, 1e-4, "lse")


def test_attention_asymmetric_exact(dev):
    """Integer-valued data where every product is exact: catches any fragment-layout permutation
    (one-hot softmax via a huge score gap picks V rows exactly)."""
    from vit_amd import ops
    B, S, H = 1, 50, 2
    E = H * 64
    q = torch.zeros(B, S, H, 64)
    k = torch.zeros(B, S, H, 64)
    v = torch.zeros(B, S, H, 64)
    perm = torch.randperm(S, generator=torch.Generator().manual_seed(1))
    for s in range(S):
        # query s matches key perm[s] only: one-hot in a 50-dim code (dims 0..49), score 64*8=512
        q[0, s, :, s] = 64.0
        k[0, perm[s], :, s] = 64.0
    v[0] = (torch.arange(S * H * 64).reshape(S, H, 64) % 127 - 63).float()
    qkv = _bf(torch.cat([q.reshape(S, E), k.reshape(S, E), v.reshape(S, E)], dim=1))
    out = torch.zeros(S, E, dtype=torch.bfloat16, device=dev)
    ops.attention(qkv.to(dev), out, B, S, H)
    ref = v[0][perm].reshape(S, E)
    assert torch.equal(out.float().cpu(), ref), "attention picks wrong V rows/columns"


@pytest.mark.parametrize("cin,cout,hw,k", [(64, 128, 56, 3), (64, 128, 56, 1), (128, 256, 28, 1), (256, 512, 14, 1)])
def test_conv_128x128_tile_bit_identical(dev, cin, cout, hw, k):
    """The 128x128 tile (tile 7: what the stride-2 convs with N % 128 == 0 run once they fill the chip) contracts K in
    the order of the 256x64 tile: same bits, so the choice by batch size cannot change a frame's result.  B = 100 is a
    batch the dispatcher sends to it by itself (tile 0)."""
    from vit_amd import ops
    B = 100
    g = torch.Generator().manual_seed(cin + hw + k)
    x = _bf(torch.randn(B, hw, hw, cin, generator=g)).to(dev)
    w = _bf(torch.randn(cout, k * k * cin, generator=g) / math.sqrt(k * k * cin)).to(dev)
    scale, bias = (torch.rand(cout, generator=g) + 0.5).to(dev), torch.randn(cout, generator=g).to(dev)
    outs = []
    for tile in (1, 7, 0):
        o = torch.empty(B, hw // 2, hw // 2, cout, dtype=torch.bfloat16, device=dev)
        ops.conv2d(x, w, o, kh=k, kw=k, stride=2, pad=k // 2, scale=scale, bias=bias, act="relu", tile=tile)
        outs.append(o)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("B,S,E,C", [(512, 50, 256, 11), (33, 197, 768, 11), (1, 50, 256, 11), (17, 197, 1024, 11),
                                     (5, 50, 128, 3), (40, 50, 192, 37), (16, 50, 384, 16)])
def test_cls_head_fused(dev, B, S, E, C):
    """Final LayerNorm of the class-token rows + Linear / GELU / Linear in one launch (csrc/head.hip) against the three
    launches it replaces: the same bits, for every width the engine sends to it; and against torch fp32."""
    from vit_amd import ops
    H = E // 2
    assert ops.cls_head_supported(E, H, C)
    g = torch.Generator().manual_seed(B + E + C)
    xs = (torch.randn(B, S * E, generator=g) * 1.7 + 0.3).to(dev)
    rows = xs[:, :E]                      # row pitch S * E, as the engine's token stream
    gam, bet = (torch.rand(E, generator=g) + 0.5).to(dev), (torch.randn(E, generator=g) * 0.2).to(dev)
    w0 = _bf(torch.randn(H, E, generator=g) / math.sqrt(E)).to(dev)
    w3 = _bf(torch.randn(C, H, generator=g) / math.sqrt(H)).to(dev)
    b0, b3 = torch.randn(H, generator=g).to(dev) * 0.1, torch.randn(C, generator=g).to(dev) * 0.1
    ln = torch.empty(B, E, dtype=torch.bfloat16, device=dev)
    h0 = torch.empty(B, H, dtype=torch.bfloat16, device=dev)
    want = torch.empty(B, C, device=dev)
    ops.layernorm(rows, gam, bet, ln, 1e-5)
    ops.gemm(ln, w0, h0, bias=b0, act="gelu")
    ops.gemm(h0, w3, want, bias=b3)
    got = torch.full((B, C), float("nan"), device=dev)
    ops.cls_head(rows, gam, bet, 1e-5, w0, b0, w3, b3, got)
    assert torch.equal(got, want), (got - want).abs().max().item()
    ref = F.gelu(F.layer_norm(rows.cpu(), (E,), gam.cpu(), bet.cpu(), 1e-5) @ w0.float().cpu().T + b0.cpu()) @ w3.float().cpu().T \
        + b3.cpu()
    _close(got, ref, 2e-2, 2e-2, "cls_head vs torch")


@pytest.mark.parametrize("B", [3, 37, 130, 256, 600])
def test_stem_conv1_training_form(dev, B):
    """rvit_stem_conv1_stats_f32_fwd (round 3): the stem's 7x7 conv of a training step read from the fp32 NCHW input by
    the rolling-band kernel, against input_pack + the implicit-GEMM conv: the stored map bit-identical (same products in
    the same kh order), the statistics equal to rvit_bn_stats of the stored map up to fp32 summation order, two runs
    identical (no atomics).  The batches cover every workgroups-per-image split (28 ... 1)."""
    from vit_amd import ops
    g = torch.Generator().manual_seed(B)
    x = (torch.randn(B, 3, 224, 224, generator=g) * 1.5).to(dev)
    w = _bf(torch.randn(64, 7 * 32, generator=g) / 12).to(dev)
    img = torch.zeros(B, 230, 232, 4, dtype=torch.bfloat16, device=dev)
    ops.input_pack(x, img, 3, 3)
    y_ref = torch.empty(B, 112, 112, 64, dtype=torch.bfloat16, device=dev)
    ops.conv2d(img, w, y_ref, kh=7, kw=1, stride=2, pad=0, seg_len=32)
    want = torch.zeros(128, device=dev)
    ops.bn_stats(y_ref, want, ops.bn_workspace(dev))
    ws = ops.conv_stats_workspace(dev)
    runs = []
    for _ in range(2):
        y = torch.full_like(y_ref, float("nan"))
        sums = torch.zeros(128, device=dev)
        ops.stem_conv1_stats_f32(x, w, y, sums, ws)
        assert torch.equal(y, y_ref), (y.float() - y_ref.float()).abs().max().item()
        runs.append(sums)
    assert torch.equal(runs[0], runs[1])
    n = B * 112 * 112
    assert torch.allclose(runs[0][:64] / n, want[:64] / n, rtol=1e-4, atol=1e-5)
    assert torch.allclose(runs[0][64:] / n, want[64:] / n, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("B", [2, 37, 130, 256, 600])
def test_stem_conv1_wgrad_roll(dev, B):
    """rvit_stem_conv1_wgrad_f32 (round 3): weight gradient of the stem's 7x7 conv from the fp32 input on the rolling band,
    against rvit_conv2d_wgrad_bf16 on the packed image (same bf16 products, another fp32 summation order) and, for the
    small batch, torch's fp32 conv weight gradient; accumulates into dw; two runs identical (ordered slots)."""
    from vit_amd import ops
    g = torch.Generator().manual_seed(100 + B)
    x = (torch.randn(B, 3, 224, 224, generator=g) * 1.5).to(dev)
    dy = _bf(torch.randn(B, 112, 112, 64, generator=g) / 64).to(dev)
    img = torch.zeros(B, 230, 232, 4, dtype=torch.bfloat16, device=dev)
    ops.input_pack(x, img, 3, 3)
    ops.enable_ordered_reductions(dev)
    want = torch.zeros(64, 224, device=dev)
    ops.conv2d_wgrad(dy, img, want, kh=7, kw=1, stride=2, pad=0, seg_len=32)
    base = torch.randn(64, 224, generator=g).to(dev)
    runs = []
    for _ in range(2):
        dw = base.clone()
        ops.stem_conv1_wgrad_f32(dy, x, dw)
        runs.append(dw - base)
    assert torch.equal(runs[0], runs[1])
    scale = want.abs().max().item()
    err = (runs[0] - want).abs().max().item()
    assert err <= 2e-4 * scale + 1e-5, (err, scale)
    if B <= 2:
        xr = x.bfloat16().float().cpu().requires_grad_(False)
        wr = torch.zeros(64, 3, 7, 7, requires_grad=True)
        F.conv2d(xr, wr, None, 2, 3).backward(dy.float().cpu().permute(0, 3, 1, 2))
        got = runs[0].view(64, 7, 8, 4)[:, :, :7, :3].permute(0, 3, 1, 2).cpu()
        _close(got, wr.grad, 1e-3, 1e-3, "stem conv1 wgrad vs torch")
    # unchanged operands
    assert torch.isfinite(runs[0]).all()


def _conv_ref(x_nhwc, w_ohwi, stride, pad):
    return F.conv2d(x_nhwc.float().permute(0, 3, 1, 2), w_ohwi.float().permute(0, 3, 1, 2), None, stride, pad)


@pytest.mark.parametrize("cin,cout,hw,k,stride,pad", [(64, 64, 56, 3, 1, 1), (64, 128, 56, 3, 2, 1),
                                                      (64, 128, 56, 1, 2, 0), (256, 512, 14, 3, 2, 1),
                                                      (512, 512, 7, 3, 1, 1), (128, 128, 28, 3, 1, 1)])
@pytest.mark.parametrize("tile", [0, 3, 7])
def test_conv_nhwc(dev, cin, cout, hw, k, stride, pad, tile):
    from vit_amd import ops
    B = 3
    g = torch.Generator().manual_seed(cin + cout + hw + k)
    x = _bf(torch.randn(B, hw, hw, cin, generator=g))
    w = _bf(torch.randn(cout, k, k, cin, generator=g) / math.sqrt(k * k * cin))
    scale = torch.rand(cout, generator=g) + 0.5
    bias = torch.randn(cout, generator=g)
    ho = (hw + 2 * pad - k) // stride + 1
    res = _bf(torch.randn(B, ho, ho, cout, generator=g))
    ref = _conv_ref(x, w, stride, pad) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)
    ref = torch.relu(ref + res.float().permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    out = torch.empty(B, ho, ho, cout, dtype=torch.bfloat16, device=dev)
    ops.conv2d(x.to(dev), w.reshape(cout, -1).to(dev), out, kh=k, kw=k, stride=stride, pad=pad,
               scale=scale.to(dev), bias=bias.to(dev), residual=res.to(dev), act="relu", tile=tile)
    _close(out, ref, 1.0 / 128, 3e-3, "conv")


def test_stem_conv1_pack_and_pool(dev):
    """fp32 NCHW -> haloed NHWC4 -> 7x7/2 conv as 7 taps of 8px x 4ch runs -> 3x3/2 max-pool."""
    from vit_amd import ops
    B, H = 2, 224
    g = torch.Generator().manual_seed(21)
    x = torch.randn(B, 3, H, H, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g) / math.sqrt(147)
    xp = torch.zeros(B, H + 6, H + 8, 4, dtype=torch.bfloat16, device=dev)
    ops.input_pack(x.to(dev), xp, 3, 3)
    ref_p = torch.zeros(B, H + 6, H + 8, 4)
    ref_p[:, 3:3 + H, 3:3 + H, :3] = _bf(x).float().permute(0, 2, 3, 1)
    assert torch.equal(xp.float().cpu(), ref_p), "input pack"
    wp = torch.zeros(64, 7, 8, 4)
    wp[:, :, :7, :3] = w.permute(0, 2, 3, 1)
    wp = _bf(wp)
    y = torch.empty(B, 112, 112, 64, dtype=torch.bfloat16, device=dev)
    ops.conv2d(xp, wp.reshape(64, 224).to(dev), y, kh=7, kw=1, stride=2, pad=0, seg_len=32, act="relu")
    ref = torch.relu(F.conv2d(_bf(x).float(), _bf(w).float(), None, 2, 3))
    _close(y, ref.permute(0, 2, 3, 1), 1.0 / 128, 3e-3, "conv1")
    p = torch.empty(B, 56, 56, 64, dtype=torch.bfloat16, device=dev)
    am = torch.empty(B, 56, 56, 64, dtype=torch.uint8, device=dev)
    ops.maxpool3x3s2(y, p, am)
    refp = F.max_pool2d(y.float().cpu().permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1)
    assert torch.equal(p.float().cpu(), refp), "maxpool"
    assert int(am.max()) <= 8


def test_embed_cls(dev):
    from vit_amd import ops
    B, S, E = 4, 50, 256
    cls = torch.randn(1, 1, E)
    pos = torch.randn(1, S, E)
    x = torch.zeros(B, S, E, device=dev)
    ops.embed_cls(cls.to(dev), pos.to(dev), x)
    assert torch.equal(x[:, 0].cpu(), (cls[0, 0] + pos[0, 0]).expand(B, E))
    assert float(x[:, 1:].abs().sum()) == 0.0


def test_errors_are_loud(dev):
    from vit_amd import ops
    from vit_amd._lib import RvitError
    a = torch.zeros(8, 12, dtype=torch.bfloat16, device=dev)   # K=12 not a multiple of 8
    w = torch.zeros(8, 12, dtype=torch.bfloat16, device=dev)
    out = torch.zeros(8, 8, dtype=torch.float32, device=dev)
    with pytest.raises(RvitError):
        ops.gemm(a, w, out)
    with pytest.raises(RuntimeError):
        ops.gemm(a.cpu(), w.cpu(), out.cpu())


@pytest.mark.parametrize("cin,cout,hw,B", [(64, 64, 56, 2), (128, 128, 28, 3), (256, 256, 14, 5), (512, 512, 7, 11),
                                            (64, 128, 14, 1), (128, 64, 7, 40), (128, 256, 28, 2), (384, 128, 14, 3)])
@pytest.mark.parametrize("plain", [False, True])
@pytest.mark.parametrize("variant", [11, 12, 13, 15, 16, 17, 18, 19])
def test_conv3x3_direct_forward_and_dgrad(dev, cin, cout, hw, B, plain, variant):
    """The LDS-resident 3x3/1 kernel (tile 10, csrc/conv3x3.hip) against torch and against the gather
    kernel (tile 1): tiles that straddle image rows and image boundaries (M = B*hw*hw is not a
    multiple of 256 for the small maps), with and without the fused epilogue, forward and flipped
    (input-gradient) form."""
    from vit_amd import ops
    if variant == 12 and (cout % 128 or cin % 128):
        pytest.skip("the 128-channel variant needs Cout (forward) and Cin (dgrad) multiples of 128")
    if variant in (15, 18) and (cout % 256 or cin % 256):
        pytest.skip("the 256-channel sixteen-wave variants need Cout and Cin multiples of 256")
    if variant in (16, 17) and hw == 56:
        pytest.skip("a 56-wide patch and a 3- / 4-stage weight ring do not fit the LDS")
    if variant == 19 and (cout % 128 or cin % 128):
        pytest.skip("the eight-wave staggered kernel needs Cout and Cin (dgrad: the roles swap) multiples of 128")
    g = torch.Generator().manual_seed(cin + cout + hw + B)
    x = _bf(torch.randn(B, hw, hw, cin, generator=g))
    w = _bf(torch.randn(cout, 3, 3, cin, generator=g) / math.sqrt(9 * cin))
    kw = {}
    ref = _conv_ref(x, w, 1, 1)
    if not plain:
        scale = torch.rand(cout, generator=g) + 0.5
        bias = torch.randn(cout, generator=g)
        res = _bf(torch.randn(B, hw, hw, cout, generator=g))
        ref = torch.relu(ref * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1) + res.float().permute(0, 3, 1, 2))
        kw = dict(scale=scale.to(dev), bias=bias.to(dev), residual=res.to(dev), act="relu")
    out = torch.empty(B, hw, hw, cout, dtype=torch.bfloat16, device=dev)
    ops.conv2d(x.to(dev), w.reshape(cout, -1).to(dev), out, kh=3, kw=3, stride=1, pad=1, tile=variant, **kw)
    _close(out, ref.permute(0, 2, 3, 1), 1.0 / 128, 3e-3, "direct conv")
    out1 = torch.empty_like(out)
    ops.conv2d(x.to(dev), w.reshape(cout, -1).to(dev), out1, kh=3, kw=3, stride=1, pad=1, tile=1, **kw)
    # same products, different fp32 summation order: at most an ulp of bf16 apart
    assert (out.float() - out1.float()).abs().max().item() <= 2.0 ** -7 * max(1.0, out1.float().abs().max().item())
    if variant >= 15:   # every direct variant contracts K in the order of the 64-channel kernel: identical bits
        ops.conv2d(x.to(dev), w.reshape(cout, -1).to(dev), out1, kh=3, kw=3, stride=1, pad=1, tile=11, **kw)
        assert torch.equal(out, out1)
    # input gradient = the same kernel on dy with flipped taps and [Cin, kh, kw, Cout] weights
    dy = _bf(torch.randn(B, hw, hw, cout, generator=g))
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    F.conv2d(xr, w.float().permute(0, 3, 1, 2), None, 1, 1).backward(dy.float().permute(0, 3, 1, 2))
    w_t = w.permute(3, 1, 2, 0).reshape(cin, 9 * cout).contiguous()
    dx = torch.empty(B, hw, hw, cin, dtype=torch.bfloat16, device=dev)
    resx = None if plain else _bf(torch.randn(B, hw, hw, cin, generator=g))
    ops.conv2d_dgrad(dy.to(dev), w_t.to(dev), dx, kh=3, kw=3, stride=1, pad=1, tile=variant,
                     residual=None if plain else resx.to(dev))
    want = xr.grad.permute(0, 2, 3, 1) + (0 if plain else resx.float())
    _close(dx, want, 1.0 / 128, 2e-2, "direct dgrad")


@pytest.mark.parametrize("c,hw,B", [(128, 28, 512), (256, 14, 512), (512, 7, 512), (256, 14, 77), (128, 28, 130),
                                     (512, 7, 1000), (256, 28, 9), (128, 14, 3)])
def test_conv3x3_eight_wave_kernel_bit_identical_at_full_size(dev, c, hw, B):
    """csrc/conv3x3_p8.hip (tile 19; what layers 2-4 of the stem run at the benchmark's batch) against the generic direct
    kernel (tile 11): BIT-EXACT (same 64-channel slice -> tap -> 32-deep MFMA step order), forward with the fused
    BatchNorm / residual / ReLU epilogue and flipped (input-gradient) form, at the bench geometry (B=512: one image,
    half an image, four images per tile: whole rounds of 256 workgroups) and at batch sizes whose tiles are ragged,
    straddle images and leave wave rows partly or wholly empty.  Launched several times: a race between the weight
    ring / patch DMA and the fragment reads of the two wave groups would show as rare differing tiles.  The automatic
    choice (tile 0) must give the same bits whichever kernel it picks."""
    from vit_amd import ops
    g = torch.Generator().manual_seed(c + hw + B)
    x = _bf(torch.randn(B, hw, hw, c, generator=g)).to(dev)
    w = _bf(torch.randn(c, 9 * c, generator=g) / math.sqrt(9 * c)).to(dev)
    res = _bf(torch.randn(B, hw, hw, c, generator=g)).to(dev)
    scale, bias = (torch.rand(c, generator=g) + 0.5).to(dev), torch.randn(c, generator=g).to(dev)
    want = torch.empty_like(x)
    ops.conv2d(x, w, want, kh=3, kw=3, stride=1, pad=1, tile=11, scale=scale, bias=bias, residual=res, act="relu")
    for rep in range(4):
        got = torch.full_like(x, float("nan"))
        ops.conv2d(x, w, got, kh=3, kw=3, stride=1, pad=1, tile=19 if rep < 3 else 0, scale=scale, bias=bias,
                   residual=res, act="relu")
        assert torch.equal(got, want), (rep, (got.float() - want.float()).abs().max().item())
    want_plain, got_plain = torch.empty_like(x), torch.full_like(x, float("nan"))
    ops.conv2d(x, w, want_plain, kh=3, kw=3, stride=1, pad=1, tile=11)
    ops.conv2d(x, w, got_plain, kh=3, kw=3, stride=1, pad=1, tile=19)
    assert torch.equal(got_plain, want_plain)
    dx_want, dx_got = torch.empty_like(x), torch.full_like(x, float("nan"))
    ops.conv2d_dgrad(x, w, dx_want, kh=3, kw=3, stride=1, pad=1, tile=11, residual=res)
    ops.conv2d_dgrad(x, w, dx_got, kh=3, kw=3, stride=1, pad=1, tile=19, residual=res)
    assert torch.equal(dx_got, dx_want)
    assert ops._conv3x3_name(B, hw, hw, c, c) in ("conv3x3_p8_256", "conv3x3_p8_128", "conv3x3_direct", "conv3x3_c256")
    if B == 512:
        assert ops._conv3x3_name(B, hw, hw, c, c).startswith("conv3x3_p8")


@pytest.mark.parametrize("B,H", [(1, 56), (3, 56), (2, 8), (37, 56), (300, 4)])
@pytest.mark.parametrize("plain", [False, True])
def test_conv3x3_weights_resident_variant(dev, B, H, plain):
    """The 64 -> 64 channel kernel with the filter bank resident in LDS (tile 14, csrc/conv3x3_c64.hip; what
    layer1 of the stem runs): against torch, and BIT-EXACT against the generic direct kernel (tile 11: same
    products, same fp32 summation order), forward and flipped (input-gradient) form; fewer tiles than CUs, more
    tiles than CUs (persistent loop, both patch buffers), single-tile images (H = 4: both halo rows zero)."""
    from vit_amd import ops
    g = torch.Generator().manual_seed(7 * B + H)
    x = _bf(torch.randn(B, H, 56, 64, generator=g))
    w = _bf(torch.randn(64, 3, 3, 64, generator=g) / 24.0)
    kw = {}
    ref = _conv_ref(x, w, 1, 1)
    if not plain:
        scale = torch.rand(64, generator=g) + 0.5
        bias = torch.randn(64, generator=g)
        res = _bf(torch.randn(B, H, 56, 64, generator=g))
        ref = torch.relu(ref * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1) + res.float().permute(0, 3, 1, 2))
        kw = dict(scale=scale.to(dev), bias=bias.to(dev), residual=res.to(dev), act="relu")
    out = torch.full((B, H, 56, 64), float("nan"), dtype=torch.bfloat16, device=dev)
    ops.conv2d(x.to(dev), w.reshape(64, -1).to(dev), out, kh=3, kw=3, stride=1, pad=1, tile=14, **kw)
    _close(out, ref.permute(0, 2, 3, 1), 1.0 / 128, 3e-3, "weights-resident conv")
    out1 = torch.empty_like(out)
    ops.conv2d(x.to(dev), w.reshape(64, -1).to(dev), out1, kh=3, kw=3, stride=1, pad=1, tile=11, **kw)
    assert torch.equal(out, out1)
    auto = torch.empty_like(out)
    ops.conv2d(x.to(dev), w.reshape(64, -1).to(dev), auto, kh=3, kw=3, stride=1, pad=1, **kw)
    assert torch.equal(out, auto)
    dy = _bf(torch.randn(B, H, 56, 64, generator=g))
    w_t = w.permute(3, 1, 2, 0).reshape(64, 9 * 64).contiguous()
    resx = None if plain else _bf(torch.randn(B, H, 56, 64, generator=g)).to(dev)
    dx = torch.full((B, H, 56, 64), float("nan"), dtype=torch.bfloat16, device=dev)
    dx1 = torch.empty_like(dx)
    ops.conv2d_dgrad(dy.to(dev), w_t.to(dev), dx, kh=3, kw=3, stride=1, pad=1, tile=14, residual=resx)
    ops.conv2d_dgrad(dy.to(dev), w_t.to(dev), dx1, kh=3, kw=3, stride=1, pad=1, tile=11, residual=resx)
    assert torch.equal(dx, dx1)
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    F.conv2d(xr, w.float().permute(0, 3, 1, 2), None, 1, 1).backward(dy.float().permute(0, 3, 1, 2))
    want = xr.grad.permute(0, 2, 3, 1) + (0 if plain else resx.float().cpu())
    _close(dx, want, 1.0 / 128, 2e-2, "weights-resident dgrad")


def test_conv3x3_weights_resident_rejects_other_geometry(dev):
    from vit_amd import ops
    from vit_amd._lib import RvitError
    x = torch.zeros(1, 28, 28, 128, dtype=torch.bfloat16, device=dev)
    w = torch.zeros(128, 9 * 128, dtype=torch.bfloat16, device=dev)
    y = torch.zeros(1, 28, 28, 128, dtype=torch.bfloat16, device=dev)
    with pytest.raises(RvitError):
        ops.conv2d(x, w, y, kh=3, kw=3, stride=1, pad=1, tile=14)


def test_conv3x3_direct_rejects_other_geometry(dev):
    from vit_amd import ops
    from vit_amd._lib import RvitError
    x = torch.zeros(1, 8, 8, 64, dtype=torch.bfloat16, device=dev)
    w = torch.zeros(64, 9 * 64, dtype=torch.bfloat16, device=dev)
    y = torch.zeros(1, 4, 4, 64, dtype=torch.bfloat16, device=dev)
    with pytest.raises(RvitError):
        ops.conv2d(x, w, y, kh=3, kw=3, stride=2, pad=1, tile=10)


@pytest.mark.parametrize("B", [1, 3])
def test_stem_conv1_pool_fused(dev, B):
    """conv7x7/2 + folded BN + ReLU + max-pool in one kernel (csrc/conv1.hip) against torch fp32 on the
    same bf16-rounded operands and against the two-kernel path (conv2d + maxpool3x3s2)."""
    from vit_amd import ops
    g = torch.Generator().manual_seed(33 + B)
    x = torch.randn(B, 3, 224, 224, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g) / math.sqrt(147)
    scale = torch.rand(64, generator=g) + 0.5
    bias = torch.randn(64, generator=g) * 0.3
    xp = torch.zeros(B, 230, 232, 4, dtype=torch.bfloat16, device=dev)
    ops.input_pack(x.to(dev), xp, 3, 3)
    wp = torch.zeros(64, 7, 8, 4)
    wp[:, :, :7, :3] = w.permute(0, 2, 3, 1)
    wp = _bf(wp).reshape(64, 224)
    w232 = torch.zeros(64, 232, dtype=torch.bfloat16)
    w232[:, :224] = wp
    pooled = torch.empty(B, 56, 56, 64, dtype=torch.bfloat16, device=dev)
    ops.stem_conv1_pool(xp, w232.to(dev), scale.to(dev), bias.to(dev), pooled)
    conv = F.conv2d(_bf(x).float(), _bf(w).float(), None, 2, 3) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)
    ref = F.max_pool2d(_bf(torch.relu(conv)).float(), 3, 2, 1).permute(0, 2, 3, 1)
    _close(pooled, ref, 1.0 / 128, 3e-3, "fused stem")
    # two-kernel path: same products in the same order -> identical up to (rare) rounding ties
    y = torch.empty(B, 112, 112, 64, dtype=torch.bfloat16, device=dev)
    ops.conv2d(xp, wp.to(dev), y, kh=7, kw=1, stride=2, pad=0, seg_len=32, scale=scale.to(dev), bias=bias.to(dev),
               act="relu")
    p2 = torch.empty_like(pooled)
    ops.maxpool3x3s2(y, p2)
    diff = (pooled.float() - p2.float()).abs()
    assert (diff > 0).float().mean().item() < 1e-3
    assert diff.max().item() <= 2.0 ** -7 * max(1.0, p2.float().abs().max().item())
    # the form inference runs: the patch built in LDS from the fp32 NCHW input itself (no packed copy) - same bits
    p3 = torch.full_like(pooled, float("nan"))
    ops.stem_conv1_pool_f32(x.to(dev), w232.to(dev), scale.to(dev), bias.to(dev), p3)
    assert torch.equal(p3, pooled)


@pytest.mark.parametrize("B", [37, 74, 130, 260, 512])
def test_stem_conv1_pool_rolling_band_splits(dev, B):
    """The rolling-band stem entry (fp32 input; what inference runs) picks 14 / 7 / 4 / 2 / 1 workgroups per image at
    these batch sizes on a 256-CU device (28 / wpi bands each, the first one computing the carried conv row on its own):
    every split must give the band kernel's bits on the packed input, including a negative folded-BN scale."""
    from vit_amd import ops
    g = torch.Generator().manual_seed(700 + B)
    x = torch.randn(B, 3, 224, 224, generator=g).to(dev)
    w = torch.randn(64, 3, 7, 7, generator=g) / math.sqrt(147)
    wp = torch.zeros(64, 7, 8, 4)
    wp[:, :, :7, :3] = w.permute(0, 2, 3, 1)
    w232 = torch.zeros(64, 232, dtype=torch.bfloat16)
    w232[:, :224] = _bf(wp).reshape(64, 224)
    w232 = w232.to(dev)
    scale = torch.rand(64, generator=g) + 0.5
    scale[5] = -0.7
    scale, bias = scale.to(dev), (torch.randn(64, generator=g) * 0.3).to(dev)
    xp = torch.zeros(B, 230, 232, 4, dtype=torch.bfloat16, device=dev)
    ops.input_pack(x, xp, 3, 3)
    ref = torch.empty(B, 56, 56, 64, dtype=torch.bfloat16, device=dev)
    ops.stem_conv1_pool(xp, w232, scale, bias, ref)
    out = torch.full_like(ref, float("nan"))
    ops.stem_conv1_pool_f32(x, w232, scale, bias, out)
    assert torch.equal(out, ref)


@pytest.mark.parametrize("name,cin,cout,hw,B,tile", [
    ("c64", 64, 64, 56, 5, 0), ("c64-many-tiles", 64, 64, 56, 24, 0), ("direct64-l4", 512, 512, 7, 37, 0),
    ("direct64-l3", 256, 256, 14, 3, 0), ("p8-128", 128, 128, 28, 256, 0), ("p8-256", 256, 256, 14, 256, 0),
    ("p8-128-ragged", 128, 128, 28, 37, 19), ("p8-256-ragged", 256, 256, 14, 37, 19), ("c256", 512, 512, 7, 512, 0),
    ("c256-ragged", 512, 512, 7, 5, 15), ("c256-l3", 256, 256, 14, 7, 15)])
def test_conv3x3_direct_batch_statistics(dev, name, cin, cout, hw, B, tile):
    """The statistics epilogue of the direct 3x3 kernels (round 3: conv3x3.hip, conv3x3_p8.hip, conv3x3_c64.hip): every
    3x3 / stride 1 conv of a training step hands its BatchNorm sums over in per-tile slots.  Stored map bit-identical to
    the plain conv's, sums equal to the exact sums of the stored map up to fp32 round-off, two runs identical."""
    from vit_amd import ops
    g = torch.Generator().manual_seed(cin + hw + B)
    x = torch.randn(B, hw, hw, cin, generator=g).to(torch.bfloat16).to(dev)
    w = _bf(torch.randn(cout, 9 * cin, generator=g) / math.sqrt(9 * cin)).to(dev)
    kw = dict(kh=3, kw=3, stride=1, pad=1, tile=tile)
    y_ref = torch.empty(B, hw, hw, cout, dtype=torch.bfloat16, device=dev)
    ops.conv2d(x, w, y_ref, **kw)
    ws = ops.conv_stats_workspace(dev)
    runs = []
    for _ in range(2):
        y = torch.full_like(y_ref, float("nan"))
        sums = torch.zeros(2 * cout, device=dev)
        fused = ops.conv2d_bn_stats(x, w, y, sums, ws, bn_workspace=ops.bn_workspace(dev), **kw)
        assert fused, f"{name}: no statistics epilogue"
        assert torch.equal(y, y_ref)
        runs.append(sums)
    assert torch.equal(runs[0], runs[1])
    yf = y_ref.float().reshape(-1, cout).double()
    exact = torch.cat([yf.sum(0), (yf ** 2).sum(0)]).float()
    scale = torch.cat([yf.abs().sum(0), (yf ** 2).sum(0)]).float().clamp_min(1e-6)
    err = ((runs[0] - exact).abs() / scale).max().item()
    assert err <= 2e-5, err


@pytest.mark.parametrize("case", ["conv1-7x7s2", "3x3s2", "1x1s2", "3x3s1-small"])
def test_conv_epilogue_batch_statistics(dev, case):
    """rvit_conv2d_fwd_stats_bf16 (round 3): where the conv kernel has the statistics epilogue, the per-tile slots summed
    by rvit_bn_stats_finalize must equal what rvit_bn_stats computes from the stored map (same bf16 values, another
    summation order: fp32 round-off), the stored map must be bit-identical to rvit_conv2d_fwd_bf16's, and two runs must
    give identical sums (no atomics)."""
    from vit_amd import ops
    g = torch.Generator().manual_seed(91)
    if case == "conv1-7x7s2":
        B = 20
        x = torch.zeros(B, 230, 232, 4, dtype=torch.bfloat16)
        x[:, 3:227, 3:227, :3] = torch.randn(B, 224, 224, 3, generator=g).to(torch.bfloat16)
        w = _bf(torch.randn(64, 7 * 32, generator=g) / 12)
        kw = dict(kh=7, kw=1, stride=2, pad=0, seg_len=32)
        out_shape = (B, 112, 112, 64)   # 980 M tiles of 256 rows: the two-stage finalisation
    elif case == "3x3s2":
        B = 6
        x = torch.randn(B, 56, 56, 64, generator=g).to(torch.bfloat16)
        w = _bf(torch.randn(128, 9 * 64, generator=g) / 24)
        kw = dict(kh=3, kw=3, stride=2, pad=1)
        out_shape = (B, 28, 28, 128)
    elif case == "1x1s2":
        B = 5
        x = torch.randn(B, 28, 28, 128, generator=g).to(torch.bfloat16)
        w = _bf(torch.randn(256, 128, generator=g) / 11)
        kw = dict(kh=1, kw=1, stride=2, pad=0)
        out_shape = (B, 14, 14, 256)
    else:
        B = 3
        x = torch.randn(B, 14, 14, 256, generator=g).to(torch.bfloat16)
        w = _bf(torch.randn(256, 9 * 256, generator=g) / 48)
        kw = dict(kh=3, kw=3, stride=1, pad=1)
        out_shape = (B, 14, 14, 256)
    x, w = x.to(dev), w.to(dev)
    C = out_shape[-1]
    y_ref = torch.empty(out_shape, dtype=torch.bfloat16, device=dev)
    ops.conv2d(x, w, y_ref, **kw)
    want = torch.zeros(2 * C, device=dev)
    ops.bn_stats(y_ref, want, ops.bn_workspace(dev))
    ws = ops.conv_stats_workspace(dev)
    runs = []
    fused = None
    for _ in range(2):
        y = torch.full(out_shape, float("nan"), dtype=torch.bfloat16, device=dev)
        sums = torch.zeros(2 * C, device=dev)
        fused = ops.conv2d_bn_stats(x, w, y, sums, ws, bn_workspace=ops.bn_workspace(dev), **kw)
        assert torch.equal(y, y_ref)
        runs.append(sums)
    assert torch.equal(runs[0], runs[1])
    yf = y_ref.float().reshape(-1, C)
    exact = torch.cat([yf.double().sum(0), (yf.double() ** 2).sum(0)]).float()
    scale = torch.cat([yf.abs().double().sum(0), (yf.double() ** 2).sum(0)]).float().clamp_min(1e-6)
    err = ((runs[0] - exact).abs() / scale).max().item()
    err_ref = ((want - exact).abs() / scale).max().item()
    print(f"\n[{case}] fused epilogue: {fused}; |sums - exact| / sum|y|: epilogue {err:.2e}, bn_stats {err_ref:.2e}")
    assert err <= 2e-5
    if case != "3x3s1-small":
        assert fused, "the implicit-GEMM kernel has the statistics epilogue"
