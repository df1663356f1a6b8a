"""Generate tests/golden/gain_golden.npz from the REAL reference (CPU, build container only):
    python tests/golden/make_golden_gain.py
``ndcg`` / ``dcg`` with a caller-supplied ``gain_function`` (allrank/models/metrics.py:7-8,41-42,67): the identity gain the
reference itself passes (losses/neuralNDCG.py:58, ``powered_relevancies=False``), a gain that is NOT zero at label 0 (padded items
then carry gain(0) at the tail positions, metrics.py:32-35,67) and a non-monotone gain (the ideal ranking is by LABEL, not by gain,
metrics.py:21)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.ref_loader import load_reference  # noqa: E402
from tests.golden.make_inputs import make_inputs  # noqa: E402

SHAPES = [(3, 7, 31, False), (5, 40, 32, False), (4, 240, 33, False), (4, 33, 34, True)]
GAINS = {"identity": lambda x: x, "plus1": lambda x: x + 1.0, "hump": lambda x: x * (3.5 - x)}
ATS = [1, 5, 10, 1000]


def build():
    load_reference(stable_sort=True)
    from allrank.models import metrics as RM
    out = {"n_cases": np.int64(len(SHAPES)), "ats": np.asarray(ATS, np.int64)}
    for ci, (B, L, seed, ties) in enumerate(SHAPES):
        s, y = make_inputs(B, L, seed, ties)
        pre = "c%d." % ci
        out[pre + "s"], out[pre + "y"] = s, y
        for name, g in GAINS.items():
            out[pre + name + ".ndcg"] = RM.ndcg(torch.tensor(s), torch.tensor(y), ats=ATS, gain_function=g).numpy()
            out[pre + name + ".dcg"] = RM.dcg(torch.tensor(s), torch.tensor(y), ats=ATS, gain_function=g).numpy()
            out[pre + name + ".ndcg_none"] = RM.ndcg(torch.tensor(s), torch.tensor(y), gain_function=g, filler_value=0.25).numpy()
    return {"gain_golden.npz": out}


def main():
    for f, d in build().items():
        np.savez_compressed(os.path.join(HERE, f), **d)
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
