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


def test_install_fit_rebinds_the_epoch_loop_with_the_reference_signature():
    """main.py:90 must reach the explicit step; here (no GPU): names and signature only"""
    import inspect
    from oracle.ref_loader import load_reference
    load_reference()
    import allrank.training.train_utils as RT
    import allrank_amd
    from allrank_amd import fit as EF
    ref_fit = RT.fit
    ref_params = list(inspect.signature(ref_fit).parameters)
    ours = list(inspect.signature(EF.fit).parameters)
    assert ours[:len(ref_params)] == ref_params, (ours, ref_params)          # same names, same order (extensions come after)
    assert all(inspect.signature(EF.fit).parameters[p].default is not inspect.Parameter.empty for p in ours[len(ref_params):])
    done = allrank_amd.install(fit=True)
    try:
        assert RT.fit is EF.fit and "allrank.training.train_utils.fit" in done
        import importlib
        main = importlib.import_module("allrank.main")      # imports `fit` from train_utils -> gets the rebound one
        assert main.fit is EF.fit
    finally:
        allrank_amd.uninstall()
    assert RT.fit is ref_fit
