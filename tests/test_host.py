"""Host-side mirror of the R interface (cocons_amd.host) against the oracle's restatement
and hand-computed values -- theta plumbing only, no GPU."""
import numpy as np
import pytest


def test_getModelLists_matches_oracle(oracle):
    import cocons_amd as ca
    rng = np.random.default_rng(0)
    pps = [
        {"mean": 0.0, "std.dev": [True] * 3, "scale": [True] * 3, "aniso": [True] * 3, "tilt": [True] * 3,
         "smooth": [True] * 3, "nugget": [True, False, False]},
        {"mean": [True, True, True], "std.dev": [True, False, True], "scale": [True, True, False],
         "aniso": 0.0, "tilt": 0.0, "smooth": 1.5, "nugget": -np.inf},
        {"mean": [True], "std.dev": [True], "scale": [True], "aniso": 0.0, "tilt": 0.0, "smooth": 0.5,
         "nugget": [True]},
    ]
    for pp in pps:
        k = sum(sum(v) for v in pp.values() if isinstance(v, list))
        th = rng.standard_normal(k)
        for ty in ("diff", "classic"):
            a, b = ca.getModelLists(th, pp, ty), oracle.getModelLists(th, pp, ty)
            assert list(a.keys()) == list(b.keys()) == list(ca.ASPECTS)
            for key in a:
                assert np.array_equal(a[key], b[key])


def test_getScale_getPen_match_oracle(oracle):
    import cocons_amd as ca
    rng = np.random.default_rng(1)
    X = np.column_stack([np.ones(30), rng.standard_normal((30, 2)) * 3 + 1])
    a, b = ca.getScale(X), oracle.getScale(X)
    for k in a:
        assert np.array_equal(a[k], b[k])
    a2 = ca.getScale(X[:5], a["mean.vector"], a["sd.vector"])
    assert np.array_equal(a2["std.covs"][:, 1], (X[:5, 1] - a["mean.vector"][1]) / a["sd.vector"][1])
    pp = {"mean": [True] * 3, "std.dev": [True] * 3, "scale": [True] * 3, "aniso": [True] * 3,
          "tilt": [True] * 3, "smooth": [True] * 3, "nugget": -np.inf}
    tl = ca.getModelLists(rng.standard_normal(18) * 0.3, pp)
    tl["tilt"][1] = 5e-5                       # exercise the smooth branch of the L1 penalty
    lam = (0.25, 0.5, 0.3)
    assert ca.getPen(100, lam, tl, (0.5, 2.5)) == pytest.approx(oracle.getPen(100, lam, tl, (0.5, 2.5)), rel=1e-15)


def test_workload_round_trip():
    import cocons_amd as ca
    from cocons_amd import workloads as wl
    th = wl.theta_full()
    pp = wl.par_pos_full()
    tv = wl.theta_vector_from_lists(th, pp)
    assert tv.size == 16
    back = ca.getModelLists(tv, pp)
    for k in ("std.dev", "scale", "aniso", "tilt", "smooth", "nugget"):
        assert np.allclose(back[k], th[k], rtol=0, atol=1e-15)
    locs = wl.grid_locs(4, 3)
    assert locs.shape == (12, 2) and locs[1, 0] > locs[0, 0] and locs[1, 1] == locs[0, 1]


def test_missing_library_fails_loudly(monkeypatch):
    from cocons_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libcocons_hip.so")
    with pytest.raises(_lib.CoconsHipError, match="no CPU fallback"):
        _lib.load()
