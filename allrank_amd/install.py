"""Drop the engine in under an UNMODIFIED allrank/main.py.

allRank resolves its plugins by name at run time (SURVEY.md §5 "Config / flags"):
    main.py:75   make_model(n_features=..., **asdict(config.model, recurse=False))     (imported name in main's namespace)
    main.py:83   getattr(allrank.models.losses, config.loss.name)
    train_utils.py:50   getattr(allrank.models.metrics, metric_name)
``install()`` rebinds exactly those attributes to the MI355X implementations; everything else of the reference
(config parsing, data loading, fit(), early stopping, logging) keeps running as is.  ``install(fit=True)`` additionally puts
``allrank_amd.fit.fit`` (reference signature, explicit HIP training step inside) behind main.py:90.  ``uninstall()`` restores all.

    import allrank_amd; allrank_amd.install(fit=True)      # then: from allrank.main import run; run()

Several GPUs: the reference wraps the model in nn.DataParallel and takes ``cuda:0`` for everybody (main.py:71-78,
models/model_utils.py:13-18,40-53, data/dataset_loading.py:240-241).  The engine's layout is one process per GPU: when a launcher
has announced a job (RANK / LOCAL_RANK / WORLD_SIZE: ``python -m allrank_amd.launch --nproc N -- <main.py args>`` or torchrun) or
a process group with more than one rank is already up, ``install()`` also binds this rank's GPU, joins the group and rebinds
``get_torch_device``, ``CustomDataParallel`` and ``create_data_loaders`` (``allrank_amd.launch``: rank-local device, identity
wrapper, global batch = world size x ``batch_size``), so the same unmodified ``main.run()`` trains slate-sharded.
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


# every module of the reference that holds its own reference to a name install(distributed) rebinds (`from x import name`)
_DEVICE_USERS = ("allrank.models.model_utils", "allrank.main", "allrank.rank_and_click", "allrank.inference.inference_utils",
                 "allrank.models.losses.neuralNDCG", "allrank.models.losses.loss_utils", "allrank.models.losses.bce",
                 "allrank.models.losses.ordinal")
_WRAPPER_USERS = ("allrank.models.model_utils", "allrank.main", "allrank.rank_and_click")
_LOADER_USERS = ("allrank.data.dataset_loading", "allrank.main")


def _install_distributed(done):
    """process-per-GPU layout (allrank_amd.launch): this rank's device instead of ``cuda:0``, no DataParallel wrapper, global
    batch = world size x batch_size.  The defining modules are imported (so later ``from ... import`` statements pick the new
    objects up); modules that already hold a copy of a name get it replaced in place."""
    from . import launch as D
    for owner in ("allrank.models.model_utils", "allrank.data.dataset_loading"):
        importlib.import_module(owner)
    for users, name, obj in ((_DEVICE_USERS, "get_torch_device", D.get_torch_device),
                             (_WRAPPER_USERS, "CustomDataParallel", D.CustomDataParallel),
                             (_LOADER_USERS, "create_data_loaders", D.create_data_loaders)):
        for modname in users:
            m = sys.modules.get(modname)
            if m is not None and hasattr(m, name):
                _set(m, name, obj)
                done.append("%s.%s" % (modname, name))


def install(losses=True, metrics=True, model=True, fit=False, distributed=None):
    """Rebind the hot-path names inside the importable ``allrank`` package.  Returns the list of rebound names.
    ``fit=True`` also rebinds the epoch loop (train_utils.py:78; imported into main.py's namespace at main.py:18) to
    ``allrank_amd.fit.fit`` -- same signature and return value, the explicit MI355X step inside.
    ``distributed``: None = when a launcher announced a multi-rank job or a multi-rank process group exists (see the module
    docstring); True / False force it.  Binding the GPU and joining the group happen here (``allrank_amd.launch.setup``)."""
    if distributed is None or distributed:
        from . import launch as D
        dev = D.setup()
        distributed = bool(distributed) or (dev is not None and D.world_size() > 1)
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
    if distributed:
        _install_distributed(done)
    return done


def uninstall():
    for (modname, name), value in list(_saved.items()):
        mod = sys.modules.get(modname)
        if mod is not None and value is not None:
            setattr(mod, name, value)
    _saved.clear()
