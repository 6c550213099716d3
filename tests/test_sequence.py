"""CPU: pins the oracle's explicit LSTM loop against torch.nn.LSTM (the reference's own module) and checks that the
product's parameter holders carry the reference's state_dict names and shapes (train_vit.py:188-318)."""
import pytest
import torch

from oracle import ref_sequence as rs

CFG = dict(backbone="patch16", embed_dim=128, num_heads=2, num_layers=1, dropout=0.0, temporal_hidden=64,
           temporal_layers=2, temporal_dropout=0.3, temporal_bidirectional=True, classifier_dropout=0.4)


@pytest.mark.parametrize("bidir,layers", [(True, 2), (False, 1), (True, 1), (False, 3)])
def test_explicit_lstm_equals_nn_lstm(bidir, layers):
    torch.manual_seed(0)
    rnn = torch.nn.LSTM(24, 16, num_layers=layers, batch_first=True, bidirectional=bidir).eval()
    x = torch.randn(3, 7, 24)
    with torch.no_grad():
        want, _ = rnn(x)
        got = rs.lstm_explicit(rnn, x)
    assert torch.allclose(got, want, atol=2e-6, rtol=1e-5)


def test_state_dict_names_and_shapes_match_reference_modules():
    from vit_amd.sequence import CNNViTSequence
    mine = CNNViTSequence(11, CFG)
    ref = rs.CNNViTSequenceRef(11, CFG)
    a = {k: tuple(v.shape) for k, v in mine.state_dict().items()}
    b = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    assert a == b
    mine.load_state_dict(ref.state_dict())                      # a reference-layout state_dict loads unchanged
    assert [n for n, _ in mine.temporal_encoder.rnn.named_parameters()] == [n for n, _ in ref.temporal_encoder.rnn.named_parameters()]


def test_gru_state_dict_matches_nn_gru():
    from vit_amd.sequence import CNNViTSequence
    cfg = dict(CFG, temporal_model="gru")
    mine, ref = CNNViTSequence(11, cfg), rs.CNNViTSequenceRef(11, cfg)
    assert {k: tuple(v.shape) for k, v in mine.state_dict().items()} == {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    mine.load_state_dict(ref.state_dict())


def test_unsupported_variants_are_loud():
    from vit_amd.sequence import CNNViTSequence, TemporalEncoder
    with pytest.raises(NotImplementedError):
        TemporalEncoder(130, hidden_dim=64)                      # input width not a multiple of 4
    with pytest.raises(ValueError):
        TemporalEncoder(128, model_type="rnn")
    m = CNNViTSequence(11, CFG)
    with pytest.raises(RuntimeError):
        m.forward_temporal_only(torch.zeros(1, 4, 128))          # CPU tensor: no fallback
