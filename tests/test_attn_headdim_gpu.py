"""Attention for head dims other than 64 (csrc/attn_dh.hip: rvit_attn_dh_fwd / rvit_attn_dh_bwd) against torch fp32.
The reference accepts any embed_dim % num_heads == 0 (train_vit.py:76-111 -> nn.MultiheadAttention); tolerances are
those of the head_dim-64 kernel's tests (tests/test_kernels_gpu.py::test_attention, test_backward_gpu.py)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

HEAD_DIMS = [16, 32, 48, 96, 128]


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _bf(t):
    return t.to(torch.bfloat16)


def _split(x, B, S, H, Dh):
    return [t.reshape(B, S, H, Dh).transpose(1, 2) for t in x.split(H * Dh, dim=1)]


@pytest.mark.parametrize("Dh", HEAD_DIMS)
@pytest.mark.parametrize("B,S,H", [(3, 50, 4), (2, 197, 3), (4, 17, 2), (1, 64, 1), (2, 130, 2), (1, 256, 2)])
def test_attention_fwd_bwd(dev, B, S, H, Dh):
    from vit_amd import ops
    E = H * Dh
    g = torch.Generator().manual_seed(B * 1000 + S + Dh)
    qkv = _bf(torch.randn(B * S, 3 * E, generator=g) * 1.2)
    dout = _bf(torch.randn(B * S, E, generator=g))
    x = qkv.float().requires_grad_(True)
    q, k, v = _split(x, B, S, H, Dh)
    sc = q @ k.transpose(-1, -2) * Dh ** -0.5
    ref = (torch.softmax(sc, -1) @ v).transpose(1, 2).reshape(B * S, E)
    ref.backward(dout.float())
    ref_lse = torch.logsumexp(sc, -1).detach()

    qd = qkv.to(dev)
    out = torch.zeros(B * S, E, dtype=torch.bfloat16, device=dev)
    lse = torch.empty(B, H, S, dtype=torch.float32, device=dev)
    ops.attention(qd, out, B, S, H, lse=lse)
    assert torch.allclose(out.float().cpu(), ref.detach(), rtol=1.0 / 64, atol=2e-2), \
        (out.float().cpu() - ref.detach()).abs().max()
    assert torch.allclose(lse.cpu(), ref_lse, rtol=1e-4, atol=1e-4)

    dqkv = torch.zeros(B * S, 3 * E, dtype=torch.bfloat16, device=dev)
    ops.attention_bwd(qd, out, dout.to(dev), lse, dqkv, B, S, H)
    for name, got, want in zip("qkv", dqkv.float().cpu().split(E, dim=1), x.grad.split(E, dim=1)):
        rel = ((got - want).norm() / want.norm()).item()
        assert rel < 0.02, f"d{name}: relative error {rel:.4f}"


@pytest.mark.parametrize("Dh", [32, 128])
def test_attention_exact_layout(dev, Dh):
    """Integer data where every product is exact: query s matches key perm[s] only (one-hot softmax through a huge
    score gap), so the output must be V's rows, permuted: catches any fragment / transposing-read permutation."""
    from vit_amd import ops
    B, S, H = 1, 90, 2
    E = H * Dh
    code = torch.randint(0, 2, (S, Dh), generator=torch.Generator().manual_seed(5)).float() * 2 - 1   # +-1 codes
    code[:, : min(Dh, 8)] = torch.tensor([[(s >> i) & 1 for i in range(min(Dh, 8))] for s in range(S)]).float() * 2 - 1
    perm = torch.randperm(S, generator=torch.Generator().manual_seed(1))
    q = torch.zeros(B, S, H, Dh)
    k = torch.zeros(B, S, H, Dh)
    for s in range(S):
        q[0, s] = code[s] * 16.0
        k[0, perm[s]] = code[s] * 16.0
    v = (torch.arange(S * H * Dh).reshape(1, S, H, Dh) % 127 - 63).float()
    qkv = _bf(torch.cat([q.reshape(S, E), k.reshape(S, E), v.reshape(S, E)], dim=1))
    out = torch.zeros(S, E, dtype=torch.bfloat16, device=dev)
    # distinct codes differ in >= 1 of the first 7-8 bits: score gap >= 2 * 256 * Dh^-0.5 >= 45 -> exp(-45) vanishes
    ops.attention(qkv.to(dev), out, B, S, H)
    assert torch.equal(out.float().cpu(), v[0][perm].reshape(S, E)), "attention picks wrong V rows / columns"


@pytest.mark.parametrize("Dh,S,H", [(32, 50, 8), (128, 197, 2), (16, 33, 3)])
def test_attention_dropout_replay(dev, Dh, S, H):
    """The dropout site is indexed as in the head_dim-64 kernels: rvit_dropout_mask replays it; forward and backward
    use the same mask."""
    from vit_amd import ops
    B, p, seed = 3, 0.2, 4242
    E = H * Dh
    g = torch.Generator().manual_seed(S + Dh)
    qb = _bf(torch.randn(B * S, 3 * E, generator=g) * 0.7).to(dev)
    db = _bf(torch.randn(B * S, E, generator=g)).to(dev)
    out = torch.empty(B * S, E, device=dev, dtype=torch.bfloat16)
    lse = torch.empty(B * H * S, device=dev)
    ops.attention(qb, out, B, S, H, lse=lse, drop_p=p, drop_seed=seed)
    dqkv = torch.empty_like(qb)
    ops.attention_bwd(qb, out, db, lse, dqkv, B, S, H, drop_p=p, drop_seed=seed)
    m = ops.attention_dropout_mask(B, H, S, p, seed, dev).float() / (1 - p)
    x = qb.float().requires_grad_(True)
    q, k, v = (u.view(B, S, H, Dh).transpose(1, 2) for u in x.chunk(3, dim=-1))
    a = torch.softmax(q @ k.transpose(-1, -2) * Dh ** -0.5, dim=-1) * m
    ref = (a @ v).transpose(1, 2).reshape(B * S, E)
    ref.backward(db.float())
    assert torch.allclose(out.float(), ref, rtol=3e-2, atol=3e-2)
    rel = ((dqkv.float() - x.grad).norm() / x.grad.norm()).item()
    assert rel < 0.02, rel


def test_unsupported_head_dim_is_refused(dev):
    from vit_amd import ops
    qkv = torch.zeros(10, 3 * 2 * 24, dtype=torch.bfloat16, device=dev)
    out = torch.zeros(10, 2 * 24, dtype=torch.bfloat16, device=dev)
    with pytest.raises(ValueError):
        ops.attention(qkv, out, 1, 10, 2)
