"""Seeded synthetic slates (SURVEY.md §8d recipe), shared by the golden generator, the parity tests and bench.py."""
import numpy as np


def make_inputs(B, L, seed, tie_scores=False):
    """N(0,1) scores, WEB30K-like label skew p=(.52,.32,.13,.02,.01), ragged lengths (slate 0 dense), slate 1 all-zero."""
    rng = np.random.default_rng(seed)
    s = rng.standard_normal((B, L)).astype(np.float32)
    if tie_scores:
        s = (np.round(s * 2) / 2).astype(np.float32)
    y = rng.choice(5, size=(B, L), p=[0.52, 0.32, 0.13, 0.02, 0.01]).astype(np.float32)
    nv = np.clip(np.round(rng.lognormal(np.log(max(L * 0.45, 1.0)), 0.6, B)), 1, L).astype(int)
    nv[0] = L
    for b in range(B):
        y[b, nv[b]:] = -1
    if B > 2:
        y[1][y[1] >= 0] = 0
    return s, y
