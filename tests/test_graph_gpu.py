"""hipGraph replay of small-batch inference (Engine._infer_graph): identical results to the eager launches, rebuilt
when the weights change, independent of the eager path's buffers."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(dev, backbone):
    from vit_amd.model import CNNViTHybrid
    from vit_amd.synth import synth_state_dict
    m = CNNViTHybrid(num_classes=11, embed_dim=128, num_heads=2, num_layers=2, dropout=0.1, backbone=backbone)
    m.load_state_dict(synth_state_dict({k: v.shape for k, v in m.state_dict().items()}, 3))
    return m.to(dev).eval()


@pytest.mark.parametrize("backbone", ["resnet18", "patch16"])
def test_graph_replay_equals_eager(dev, backbone):
    from vit_amd.synth import micro_doppler_batch, synth_state_dict
    m = _model(dev, backbone)
    xs = [micro_doppler_batch(B, seed=10 + B).to(dev) for B in (1, 3, 1, 5)]
    with torch.no_grad():
        eager = [m(x) for x in xs]
        eager_f = m.forward_features(xs[0])
        m.set_graph_capture(4)
        for _ in range(2):                                   # first pass captures, second replays
            got = [m(x) for x in xs]                         # B=5 exceeds max_batch: eager
            for a, b in zip(eager, got):
                assert torch.equal(a, b)
            assert torch.equal(m.forward_features(xs[0]), eager_f)
        assert len(m.engine._pack["_graphs"]) == 3           # (1, logits), (3, logits), (1, features)
        # outputs are copies: a later replay does not overwrite an earlier result
        r1 = m(xs[0]); keep = r1.clone(); m(xs[2]); assert torch.equal(r1, keep)
        # new weights -> new pack -> new graphs
        m.load_state_dict(synth_state_dict({k: v.shape for k, v in m.state_dict().items()}, 4))
        fresh = _model(dev, backbone)
        fresh.load_state_dict(m.state_dict())
        assert torch.equal(m(xs[0]), fresh(xs[0]))
        assert not torch.equal(m(xs[0]), eager[0])
        m.set_graph_capture(0)
        assert torch.equal(m(xs[0]), fresh(xs[0]))
