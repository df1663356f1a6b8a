"""CPU checks of the model plugin surface: same parameter names/shapes/count as the reference and -- when the
reference tree is present (build container only) -- bit-identical initial weights under the same seed."""
import pytest
import torch

from oracle.ref_loader import reference_available


def _engine(cfgkw):
    from allrank_amd.model import make_model
    return make_model(**cfgkw)


def _kw(N=2, h=8, d_ff=2048, sizes=(512,), nf=136, act=None, norm=False, oact=None, pos=None):
    return dict(fc_model=dict(sizes=list(sizes), input_norm=norm, activation=act, dropout=0.0),
                transformer=dict(N=N, d_ff=d_ff, h=h, positional_encoding=pos, dropout=0.0) if N else None,
                post_model=dict(d_output=1, output_activation=oact), n_features=nf)


def test_state_dict_contract_config3():
    m = _engine(_kw())
    sd = m.state_dict()
    assert sum(v.numel() for v in sd.values()) == 6376449          # SURVEY.md §8a row a2 [probe]
    for n in range(2):
        for j in range(4):
            assert sd["encoder.layers.%d.self_attn.linears.%d.weight" % (n, j)].shape == (512, 512)
        assert sd["encoder.layers.%d.feed_forward.w_1.weight" % n].shape == (2048, 512)
        assert sd["encoder.layers.%d.sublayer.1.norm.a_2" % n].shape == (512,)
    assert sd["input_layer.layers.0.weight"].shape == (512, 136) and sd["encoder.norm.b_2"].shape == (512,)
    assert sd["output_layer.w_1.weight"].shape == (1, 512)


def test_make_model_mutates_sizes_like_reference():
    kw = _kw(N=0, sizes=(16,), nf=20)
    _engine(kw)
    assert kw["fc_model"]["sizes"] == [20, 16]                     # model.py:25


@pytest.mark.skipif(not reference_available(), reason="reference tree only exists in the build container")
@pytest.mark.parametrize("kw", [_kw(N=2, h=4, d_ff=64, sizes=(32,), nf=20), _kw(N=0, sizes=(24, 16), nf=20, act="ReLU", norm=True),
                                _kw(N=1, h=1, d_ff=48, sizes=(24,), nf=12, oact="Sigmoid"),
                                _kw(N=1, h=2, d_ff=48, sizes=(24,), nf=12, pos=dict(strategy="learned", max_indices=30)),
                                _kw(N=1, h=2, d_ff=48, sizes=(24,), nf=12, pos=dict(strategy="fixed", max_indices=30))])
def test_same_seed_gives_reference_initial_weights(kw):
    import copy
    from oracle.ref_loader import load_reference
    load_reference()
    from allrank.models.model import make_model as ref_make
    from allrank.config import TransformerConfig
    k1, k2 = copy.deepcopy(kw), copy.deepcopy(kw)
    if k1["transformer"]:
        from allrank.config import PositionalEncoding
        if k1["transformer"]["positional_encoding"]:
            k1["transformer"]["positional_encoding"] = PositionalEncoding(**k1["transformer"]["positional_encoding"])
        k1["transformer"] = TransformerConfig(**k1["transformer"])
    torch.manual_seed(42)
    ref = ref_make(**k1)
    torch.manual_seed(42)
    eng = _engine(k2)
    rs, es = ref.state_dict(), eng.state_dict()
    assert list(rs.keys()) == list(es.keys())
    for k in rs:
        assert torch.equal(rs[k], es[k]), k
    assert [n for n, _ in ref.named_parameters()] == [n for n, _ in eng.named_parameters()]
    if kw["transformer"] and kw["transformer"]["positional_encoding"]:      # the position module computes the same thing
        x = torch.randn(3, 7, kw["fc_model"]["sizes"][-1])
        idx = torch.tensor([[0, 1, 2, 3, 4, 5, 6], [5, 40, 2, -1, -1, -1, -1], [29, 30, 31, 0, 1, -1, -1]])
        mask = idx == -1
        assert torch.equal(ref.encoder.position(x, mask, idx.clone()), eng.encoder.position(x, mask, idx.clone()))
