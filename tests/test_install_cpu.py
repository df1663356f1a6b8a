"""install() rebinds exactly the names allrank/main.py looks up (build container only: needs /root/reference)."""
import pytest

from oracle.ref_loader import reference_available

pytestmark = pytest.mark.skipif(not reference_available(), reason="reference tree only exists in the build container")


def test_install_rebinds_hot_path_names_and_uninstall_restores():
    from oracle.ref_loader import load_reference
    load_reference()
    import allrank.models.losses as RL
    import allrank.models.metrics as RM
    import allrank.models.model as RMod
    import allrank_amd
    from allrank_amd import losses as E, metrics as EM, model as EMod
    orig = (RL.approxNDCGLoss, RM.ndcg, RMod.make_model, RL.rankNet)
    done = allrank_amd.install()
    try:
        assert RL.approxNDCGLoss is E.approxNDCGLoss and RL.listNet is E.listNet and RL.lambdaLoss is E.lambdaLoss
        assert RL.neuralNDCG is E.neuralNDCG and RL.listMLE is E.listMLE
        assert RM.ndcg is EM.ndcg and RMod.make_model is EMod.make_model
        assert RL.rankNet is E.rankNet and RL.ordinal is E.ordinal and RL.bce is E.bce and RM.mrr is EM.mrr     # §8f row 4
        assert RL.with_ordinals is not None
        assert getattr(RL, "approxNDCGLoss") is E.approxNDCGLoss      # what main.py:83 does
        assert len(done) >= 9
    finally:
        allrank_amd.uninstall()
    assert (RL.approxNDCGLoss, RM.ndcg, RMod.make_model, RL.rankNet) == orig
