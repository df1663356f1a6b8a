"""Drop the engine in under an UNMODIFIED allrank/main.py.

allRank resolves its plugins by name at run time (SURVEY.md §5 "Config / flags"):
    main.py:75   make_model(n_features=..., **asdict(config.model, recurse=False))     (imported name in main's namespace)
    main.py:83   getattr(allrank.models.losses, config.loss.name)
    train_utils.py:50   getattr(allrank.models.metrics, metric_name)
``install()`` rebinds exactly those attributes to the MI355X implementations; everything else of the reference
(config parsing, data loading, fit(), early stopping, logging) keeps running as is.  ``install(fit=True)`` additionally puts
``allrank_amd.fit.fit`` (reference signature, explicit HIP training step inside) behind main.py:90.  ``uninstall()`` restores all.

    import allrank_amd; allrank_amd.install(fit=True)      # then: from allrank.main import run; run()
"""
import importlib
import sys

_HOT_LOSSES = ("listNet", "listMLE", "approxNDCGLoss", "lambdaLoss", "neuralNDCG", "neuralNDCG_transposed",
               "rankNet", "rankNet_weightByGTDiff", "rankNet_weightByGTDiff_pow", "bce", "ordinal", "pointwise_rmse",
               "binary_listNet")
_HOT_METRICS = ("ndcg", "dcg", "mrr")
_saved = {}


def _set(mod, name, value):
    _saved.setdefault((mod.__name__, name), getattr(mod, name, None))
    setattr(mod, name, value)


def install(losses=True, metrics=True, model=True, fit=False):
    """Rebind the hot-path names inside the importable ``allrank`` package.  Returns the list of rebound names.
    ``fit=True`` also rebinds the epoch loop (train_utils.py:78; imported into main.py's namespace at main.py:18) to
    ``allrank_amd.fit.fit`` -- same signature and return value, the explicit MI355X step inside."""
    from . import losses as E, metrics as EM, model as EMod
    done = []
    if losses:
        rl = importlib.import_module("allrank.models.losses")
        for n in _HOT_LOSSES:
            _set(rl, n, getattr(E, n))
            done.append("allrank.models.losses." + n)
    if metrics:
        rm = importlib.import_module("allrank.models.metrics")
        for n in _HOT_METRICS:
            _set(rm, n, getattr(EM, n))
            done.append("allrank.models.metrics." + n)
    if model:
        rmod = importlib.import_module("allrank.models.model")
        _set(rmod, "make_model", EMod.make_model)
        done.append("allrank.models.model.make_model")
        for modname in ("allrank.main", "allrank.rank_and_click"):      # `from allrank.models.model import make_model`
            m = sys.modules.get(modname)
            if m is not None and hasattr(m, "make_model"):
                _set(m, "make_model", EMod.make_model)
                done.append(modname + ".make_model")
    if fit:
        from . import fit as EF
        rt = importlib.import_module("allrank.training.train_utils")
        _set(rt, "fit", EF.fit)
        done.append("allrank.training.train_utils.fit")
        m = sys.modules.get("allrank.main")                         # `from allrank.training.train_utils import fit`
        if m is not None and hasattr(m, "fit"):
            _set(m, "fit", EF.fit)
            done.append("allrank.main.fit")
    return done


def uninstall():
    for (modname, name), value in list(_saved.items()):
        mod = sys.modules.get(modname)
        if mod is not None and value is not None:
            setattr(mod, name, value)
    _saved.clear()
