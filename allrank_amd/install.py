"""Drop the engine in under an UNMODIFIED allrank/main.py.

allRank resolves its plugins by name at run time (SURVEY.md §5 "Config / flags"):
    main.py:57   load_libsvm_dataset(input_path=..., slate_length=..., validation_ds_role=...)   (imported name in main's namespace)
    main.py:67   create_data_loaders(train_ds, val_ds, num_workers=..., batch_size=...)          (imported name in main's namespace)
    main.py:75   make_model(n_features=..., **asdict(config.model, recurse=False))     (imported name in main's namespace)
    main.py:83   getattr(allrank.models.losses, config.loss.name)
    train_utils.py:50   getattr(allrank.models.metrics, metric_name)
``install()`` rebinds exactly those attributes to the MI355X implementations; everything else of the reference
(config parsing, early stopping, logging) keeps running as is.  ``install(fit=True)`` additionally puts
``allrank_amd.fit.fit`` (reference signature, explicit HIP training step inside) behind main.py:90.  ``uninstall()`` restores all.

    import allrank_amd; allrank_amd.install(fit=True)      # then: from allrank.main import run; run()

Data (round 6): with a GPU present ``install()`` also rebinds ``load_libsvm_dataset`` / ``create_data_loaders``
(dataset_loading.py:197-248, main.py:8) to the device-resident loader of ``allrank_amd.data`` -- the libsvm files are parsed on the GPU
once, every batch is assembled in HBM (FixLength on the device) in the reference loader's batch order, nothing crosses PCIe per step.
``install(data=False)`` keeps the reference's host DataLoader (1-2 M slots/s on 8 cores against a 7 M items/s step).

Several GPUs: the reference wraps the model in nn.DataParallel and takes ``cuda:0`` for everybody (main.py:71-78,
models/model_utils.py:13-18,40-53, data/dataset_loading.py:240-241).  The engine's layout is one process per GPU
(``python -m allrank_amd.launch --nproc N -- <main.py args>`` or torchrun).  ``install()`` itself never creates a process group
(ADVICE r5): it ADOPTS one that is already initialised with more than one rank -- ``allrank_amd.launch`` initialises it before it
calls install(); a host program may have initialised its own -- and then also rebinds ``get_torch_device``, ``CustomDataParallel`` and
``create_data_loaders`` (rank-local device, identity wrapper, global batch = world size x ``batch_size``, each rank assembling only its
block of it).  ``install(distributed=True)`` is the explicit request to bind this rank's GPU and join the group the environment
announces (RANK / LOCAL_RANK / WORLD_SIZE) right here.
"""
import importlib
import logging
import sys

log = logging.getLogger("allrank_amd.install")

_HOT_LOSSES = ("listNet", "listMLE", "approxNDCGLoss", "lambdaLoss", "neuralNDCG", "neuralNDCG_transposed",
               "rankNet", "rankNet_weightByGTDiff", "rankNet_weightByGTDiff_pow", "bce", "ordinal", "pointwise_rmse",
               "binary_listNet")
_HOT_METRICS = ("ndcg", "dcg", "mrr")
_saved = {}
_ours = set()            # id() of every object install() has bound: never recorded as somebody's "original"


def _set(mod, name, value):
    """bind ``mod.name = value``; the FIRST foreign object seen under that name is what uninstall() restores.  A module that
    imported the name after an earlier install() (``from x import name`` executed once x was already rebound -- allrank.main under
    ``launch.run_main``) holds OUR object: its original is then the defining module's saved original, not our own replacement
    (uninstall() used to 'restore' allrank.main to the engine versions)."""
    key = (mod.__name__, name)
    cur = getattr(mod, name, None)
    if key not in _saved:
        if id(cur) in _ours:
            cur = next((v for (m, n), v in _saved.items() if n == name and v is not None and id(v) not in _ours), None)
        _saved[key] = cur
    _ours.add(id(value))
    setattr(mod, name, value)


# every module of the reference that holds its own reference to a name install(distributed) rebinds (`from x import name`)
_DEVICE_USERS = ("allrank.models.model_utils", "allrank.main", "allrank.rank_and_click", "allrank.inference.inference_utils",
                 "allrank.models.losses.neuralNDCG", "allrank.models.losses.loss_utils", "allrank.models.losses.bce",
                 "allrank.models.losses.ordinal")
_WRAPPER_USERS = ("allrank.models.model_utils", "allrank.main", "allrank.rank_and_click")
_LOADER_USERS = ("allrank.data.dataset_loading", "allrank.main")


def _rebind(users, name, obj, done):
    for modname in users:
        m = sys.modules.get(modname)
        if m is not None and hasattr(m, name):
            _set(m, name, obj)
            done.append("%s.%s" % (modname, name))


def _install_distributed(done):
    """process-per-GPU layout (allrank_amd.launch): this rank's device instead of ``cuda:0``, no DataParallel wrapper, global
    batch = world size x batch_size.  The defining modules are imported (so later ``from ... import`` statements pick the new
    objects up); modules that already hold a copy of a name get it replaced in place."""
    from . import launch as D
    for owner in ("allrank.models.model_utils", "allrank.data.dataset_loading"):
        importlib.import_module(owner)
    _rebind(_DEVICE_USERS, "get_torch_device", D.get_torch_device, done)
    _rebind(_WRAPPER_USERS, "CustomDataParallel", D.CustomDataParallel, done)
    _rebind(_LOADER_USERS, "create_data_loaders", D.create_data_loaders, done)


def _install_data(done):
    """the two loader names main.py imports (main.py:8) -> the device-resident loader.  ``load_libsvm_dataset_role`` keeps the
    reference's host dataset: its other caller, rank_and_click.py:63, feeds a host DataLoader with worker processes."""
    from . import data as ED
    importlib.import_module("allrank.data.dataset_loading")
    _rebind(_LOADER_USERS, "load_libsvm_dataset", ED.load_libsvm_dataset, done)
    _rebind(_LOADER_USERS, "create_data_loaders", ED.create_data_loaders, done)


def install(losses=True, metrics=True, model=True, fit=False, distributed=None, data=None):
    """Rebind the hot-path names inside the importable ``allrank`` package.  Returns the list of rebound names.
    ``fit=True`` also rebinds the epoch loop (train_utils.py:78; imported into main.py's namespace at main.py:18) to
    ``allrank_amd.fit.fit`` -- same signature and return value, the explicit MI355X step inside.
    ``data``: None = rebind the dataset / loader names to the device-resident loader when a GPU is present; True / False force it.
    ``distributed``: None = adopt a process group that is ALREADY initialised with more than one rank (nothing is initialised here);
    True = bind this rank's GPU and join the group the launcher environment announces (``allrank_amd.launch.setup``); False = never."""
    from . import launch as D
    if distributed:
        dev = D.setup()
        distributed = dev is not None and D.world_size() > 1
    elif distributed is None:
        dev = D.setup(init=False)
        distributed = dev is not None and D.world_size() > 1
        if not distributed and D.distributed_env() is not None:
            log.warning("allrank_amd.install: the environment announces rank %s of %s but no torch.distributed process group is "
                        "initialised -- install() does not create one.  Start the job through `python -m allrank_amd.launch`, call "
                        "allrank_amd.launch.setup() / install(distributed=True), or initialise the group before install(); until "
                        "then this process trains alone.", *D.distributed_env()[::2])
    if data is None:
        import torch
        data = torch.cuda.is_available()
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
        _rebind(("allrank.main", "allrank.rank_and_click"), "make_model", EMod.make_model, done)   # `from allrank.models.model import make_model`
    if fit:
        from . import fit as EF
        rt = importlib.import_module("allrank.training.train_utils")
        _set(rt, "fit", EF.fit)
        done.append("allrank.training.train_utils.fit")
        _rebind(("allrank.main",), "fit", EF.fit, done)                 # `from allrank.training.train_utils import fit`
    if distributed:
        _install_distributed(done)
    if data:
        _install_data(done)
    return done


def uninstall():
    for (modname, name), value in list(_saved.items()):
        mod = sys.modules.get(modname)
        if mod is not None and value is not None:
            setattr(mod, name, value)
    _saved.clear()
    _ours.clear()
