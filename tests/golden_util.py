"""Helpers for reading tests/golden/*.npz (written by oracle/make_golden.py from the reference itself)."""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def sub(fx, name, arr):
    """subsample `arr` the way the fixture `name` was packed; returns (golden_subsample, arr_subsample)"""
    a = np.asarray(arr, dtype=np.float32).reshape(-1)
    stride = int(fx[name + ".__stride"])
    return fx[name], a[::stride]


def check_packed(fx, name, arr, atol, rtol, norm_rtol=None):
    g, a = sub(fx, name, arr)
    assert g.shape == a.shape, f"{name}: shape {a.shape} vs golden {g.shape}"
    err = np.abs(g - a)
    tol = atol + rtol * np.abs(g)
    assert (err <= tol).all(), f"{name}: max err {err.max():.3e} (tol {tol[err.argmax()]:.3e}) at {err.argmax()}"
    if norm_rtol is not None:
        n = float(np.sqrt((np.asarray(arr, dtype=np.float64) ** 2).sum()))
        gn = float(fx[name + ".__norm"])
        assert abs(n - gn) <= norm_rtol * max(gn, 1e-12), f"{name}: norm {n} vs golden {gn}"


def cosine(a, b):
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    return float((a * b).sum() / (np.sqrt((a * a).sum() * (b * b).sum()) + 1e-30))
