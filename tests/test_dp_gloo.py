"""CPU-only, world_size 2 over gloo: the bucketed gradient exchange used by the training path
(vit_amd/dp.py + the flat gradient layout of vit_amd/trainer.py).  No kernel runs here; the test
covers what must be right by construction for N>1: the layout covers every parameter exactly once
in backward-completion order, buckets tile the flat buffer without gaps or overlap, and after
`finish` every rank holds the rank-average (N GPUs x B == 1 GPU x N*B up to summation order,
SURVEY.md §8e)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, bucket_bytes, marks, out):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vit_amd.dp import BucketReducer
        red = BucketReducer(total, None, bucket_bytes)
        results = []
        for step in range(2):  # two steps: the reducer must reset between them
            g = torch.Generator().manual_seed(100 * step + rank)
            flat = torch.randn(total, generator=g)
            red.begin_step()
            for upto in marks:  # the backward pass reports finished prefixes
                red.ready(flat, upto)
            red.finish(flat)
            results.append((flat.clone(), list(red.buckets)))
        out[rank] = results
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_averages_and_tiles():
    world, total, bucket_bytes = 2, 10_000, 4 * 3000
    marks = [640, 2560, 2600, 5200, 9984]
    mgr = mp.Manager()
    out = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(world, port, total, bucket_bytes, marks, out), nprocs=world, join=True)
    for step in range(2):
        want = sum(torch.randn(total, generator=torch.Generator().manual_seed(100 * step + r)) for r in range(world))
        want = want / world
        for r in range(world):
            flat, buckets = out[r][step]
            assert torch.allclose(flat, want, atol=1e-6), f"rank {r} step {step}"
            # buckets tile [0,total) exactly, each but the last at least bucket-sized
            assert buckets[0][0] == 0 and buckets[-1][1] == total
            for (s0, e0), (s1, e1) in zip(buckets, buckets[1:]):
                assert e0 == s1
            assert all((e - s) * 4 >= bucket_bytes for s, e in buckets[:-1])
            assert len(buckets) >= 2, "the exchange must be split so it can overlap with backward"


def test_grad_layout_covers_every_parameter_once():
    from vit_amd.model import CNNViTHybrid
    from vit_amd.trainer import _GradLayout
    for cfg in (dict(num_classes=11, embed_dim=256, num_heads=4, num_layers=4, dropout=0.0, backbone="resnet18"),
                dict(num_classes=11, embed_dim=128, num_heads=2, num_layers=2, dropout=0.0, backbone="patch16")):
        m = CNNViTHybrid(**cfg)
        lay = _GradLayout(m)
        names = dict(m.named_parameters())
        assert set(lay.names) == set(names) and len(lay.names) == len(names)
        spans = sorted((off, off + n) for off, n, _ in lay.offsets.values())
        assert spans[0][0] == 0 and all(a[1] <= b[0] for a, b in zip(spans, spans[1:])) and spans[-1][1] <= lay.total
        assert all(off % 64 == 0 for off, _, _ in lay.offsets.values())
        # backward-completion order: the head first, the first stem conv (or the patch conv) last
        assert lay.names[0].startswith("head.3")
        assert lay.names[-1] == ("cnn_backbone.conv1.weight" if cfg["backbone"] != "patch16" else "patch_embed.bias")
        assert lay.param_order == [n for n, _ in m.named_parameters()]
        if cfg["backbone"] != "patch16":
            # the backward pass reports a stem block as final through its conv1.weight: every parameter of the
            # block, and of every later layer, must lie at or before it in the flat buffer
            off = {n: o for n, (o, _, _) in lay.offsets.items()}
            blocks = sorted({n.rsplit(".", 2)[0] + "." for n in off if ".layer" in n and ".conv1.weight" in n})
            for pref in blocks:
                mine = [o for n, o in off.items() if n.startswith(pref)]
                assert max(mine) == off[pref + "conv1.weight"], pref
            order = [f"cnn_backbone.layer{li}.{bi}." for li in (4, 3, 2, 1) for bi in (1, 0)]
            closes = [off[p + "conv1.weight"] for p in order]
            assert closes == sorted(closes)


def test_reducer_requires_initialised_process_group():
    from vit_amd.dp import BucketReducer
    with pytest.raises(RuntimeError):
        BucketReducer(100)
