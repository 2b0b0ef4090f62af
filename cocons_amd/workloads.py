"""Synthetic inputs of the BASELINE configurations, shared by
bench.py and the tests so that both measure and check the same workloads."""
from __future__ import annotations

from collections import OrderedDict

import numpy as np

from .host import getScale


def grid_locs(gx: int, gy: int | None = None) -> np.ndarray:
    """expand.grid(seq(0,1,len=gx), seq(0,1,len=gy)) -- first coordinate varies fastest."""
    gy = gx if gy is None else gy
    xs, ys = np.linspace(0.0, 1.0, gx), np.linspace(0.0, 1.0, gy)
    return np.column_stack([np.tile(xs, gy), np.repeat(ys, gx)])


def design_from_locs(locs: np.ndarray, mean_vector=None, sd_vector=None):
    """X = [1, cov_x, cov_y] with cov_x = x, cov_y = y, standardised exactly as getScale
    (R/getFunctions.R:410-434)."""
    X = np.column_stack([np.ones(locs.shape[0]), locs[:, 0], locs[:, 1]])
    return getScale(X, mean_vector, sd_vector)


def theta_full(nugget=True, scale0=np.log(0.05)) -> OrderedDict:
    """The well-conditioned full nonstationary parameter set of SURVEY §8d (the 6 x p table
    the kernel sees, i.e. after getModelLists), mean = 0."""
    th = OrderedDict()
    th["mean"] = np.zeros(3)
    th["std.dev"] = np.array([0.0, 0.3, -0.2])
    th["scale"] = np.array([float(scale0), 0.2, 0.1])
    th["aniso"] = np.array([0.0, 0.25, -0.25])
    th["tilt"] = np.array([0.0, 0.3, 0.3])
    th["smooth"] = np.array([0.0, 0.5, -0.5])
    th["nugget"] = np.array([np.log(1e-2), 0.0, 0.0]) if nugget else np.array([-np.inf, 0.0, 0.0])
    return th


SMOOTH_LIMITS = (0.5, 2.5)


def par_pos_full(p=3) -> OrderedDict:
    """par.pos of the C4 optimisation: 5 covariance aspects x p columns free + nugget
    intercept free, mean fixed at 0 (P = 5p + 1 free parameters)."""
    pp = OrderedDict()
    pp["mean"] = 0.0
    for k in ("std.dev", "scale", "aniso", "tilt", "smooth"):
        pp[k] = [True] * p
    pp["nugget"] = [True] + [False] * (p - 1)
    return pp


def theta_vector_from_lists(theta_list, par_pos) -> np.ndarray:
    """Inverse of getModelLists(type='diff') for par_pos_full: returns the optimiser vector."""
    sd, sc = np.array(theta_list["std.dev"], float), np.array(theta_list["scale"], float)
    raw = {k: np.array(v, float) for k, v in theta_list.items()}
    raw["std.dev"], raw["scale"] = sd + sc, sd - sc          # sd' = (a+b)/2, sc' = (a-b)/2
    out = []
    for k, pp in par_pos.items():
        if isinstance(pp, (list, tuple)):
            out.extend(raw[k][np.asarray(pp, bool)].tolist())
    return np.asarray(out)


def synthetic_z(n: int, seed=20251114) -> np.ndarray:
    """One realization column.  (For the benchmark the data only need to be fixed and
    reproducible; an i.i.d. N(0,1) vector keeps the quadratic form well scaled.)"""
    return np.random.default_rng(seed).standard_normal(n)
