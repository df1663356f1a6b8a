"""allrank_amd -- MI355X (gfx950) native engine for allRank's listwise-LTR training hot path.

Drop-in surface (same names/signatures as the reference):
    allrank_amd.losses   <->  allrank.models.losses     (listNet, listMLE, approxNDCGLoss, neuralNDCG*, lambdaLoss)
    allrank_amd.metrics  <->  allrank.models.metrics    (ndcg, dcg)
    allrank_amd.model    <->  allrank.models.model      (make_model -> LTRModel with forward/score)
    allrank_amd.install()     rebinds those names inside an importable ``allrank`` package so that an unmodified
                              allrank/main.py picks the engine up through its getattr lookups.
Behind it: libltrx.so, a C-ABI library of hand-written HIP kernels (include/ltrx.h, allrank_amd/csrc/).
"""
__version__ = "0.1.0"

from . import losses, metrics  # noqa: F401,E402
from .install import install, uninstall  # noqa: F401,E402
