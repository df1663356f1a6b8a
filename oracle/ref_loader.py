"""Import the *real* allRank reference (read-only, /root/reference) on CPU  --  TEST INFRASTRUCTURE.

Exists only to (a) validate oracle/ltr_oracle.py against the reference itself in the build container
and (b) generate the golden vectors under tests/golden/.  /root/reference does not exist on the GPU
box, so nothing that runs there (``-m gpu`` tests, smoke(), bench.py) may call ``load_reference``.

The reference pulls in packages that are not installed here: torchvision
(dataset_loading.py:8-9), gcsfs (file_utils.py:6), tensorboardX, flatten_dict.  They are replaced by
in-memory stub modules; nothing is written to the reference tree (PYTHONDONTWRITEBYTECODE).
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("ALLRANK_REFERENCE", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "allrank"))


def _stub(name, **attrs):
    if name in sys.modules:
        return sys.modules[name]
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def load_reference(stable_sort=True):
    """Returns the imported ``allrank`` package of the reference.  With ``stable_sort`` the reference's
    ``Tensor.sort`` calls run with stable=True (the tie policy of SURVEY.md §9.2)."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    sys.dont_write_bytecode = True
    import torch

    class _Compose(object):
        def __init__(self, transforms):
            self.transforms = transforms

        def __call__(self, x):
            for t in self.transforms:
                x = t(x)
            return x

    try:
        import torchvision  # noqa: F401
    except Exception:
        tv = _stub("torchvision")
        tvt = _stub("torchvision.transforms", Compose=_Compose)
        tv.transforms = tvt
    try:
        import gcsfs  # noqa: F401
    except Exception:
        _stub("gcsfs", GCSFileSystem=type("GCSFileSystem", (), {}))
    try:
        import tensorboardX  # noqa: F401
    except Exception:
        class _SW(object):
            def __init__(self, *a, **k):
                pass

            def add_scalar(self, *a, **k):
                pass

            def close(self):
                pass
        _stub("tensorboardX", SummaryWriter=_SW)
    try:
        import flatten_dict  # noqa: F401
    except Exception:
        def _flatten(d, reducer="path", parent=()):
            out = {}
            for k, v in d.items():
                key = parent + (str(k),)
                if isinstance(v, dict):
                    out.update(_flatten(v, reducer, key))
                else:
                    out["/".join(key)] = v
            return out
        _stub("flatten_dict", flatten=_flatten)
        _stub("flatten_dict.reducers", make_reducer=lambda delimiter="/": "path")
    if REFERENCE_ROOT not in sys.path:
        sys.path.append(REFERENCE_ROOT)       # appended, not prepended: the reference has its own `tests` package
    import allrank  # noqa: E402
    import allrank.models.losses  # noqa: F401,E402
    import allrank.models.metrics  # noqa: F401,E402
    import allrank.models.model  # noqa: F401,E402
    if stable_sort and not getattr(torch.Tensor.sort, "_ltrx_stable", False):
        _orig = torch.Tensor.sort

        def _stable_sort(self, *args, **kwargs):
            kwargs.setdefault("stable", True)
            return _orig(self, *args, **kwargs)
        _stable_sort._ltrx_stable = True
        torch.Tensor.sort = _stable_sort
    return allrank
