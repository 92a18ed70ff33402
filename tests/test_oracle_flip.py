"""CPU: the decision-flip allowance of the oracle's backward (gs_oracle.cpp, render_backward; used by the GPU parity
tests) is checked here against an independent "other implementation": the same oracle in fp64.  fp32 and fp64 take a
different branch at a few (pixel, splat) pairs that sit on a blend threshold; the nine pixel-stage sums of the two must
agree within 1e-4 * sum|term| + flip9(margin) for EVERY splat once the margin covers fp32 rounding, and the allowance
must be what closes the gap (in the dense scene some splats are out of budget without it)."""
import numpy as np

from util import make_scene, view_parts

NINE = {"dL_dcolor": ([0, 1, 2], 3, [0, 1, 2]), "dL_dmean2D": ([3, 4], 3, [0, 1]), "dL_dconic": ([5, 6, 7], 4, [0, 1, 3]),
        "dL_dopacity": ([8], 1, [0])}


def _bad(P, g, want, abs9, flip9):
    bad = np.zeros(P, bool)
    for name, (qs, stride, cols) in NINE.items():
        a, b = g[name].reshape(P, stride), want[name].reshape(P, stride)
        for q, c in zip(qs, cols):
            tol = 1e-4 * np.maximum(abs9[:, q], 1e-3 * abs9[:, q].max() + 1e-30) + flip9[:, q]
            bad |= np.abs(a[:, c].astype(np.float64) - b[:, c]) > tol
    return bad


def test_flip_allowance_explains_fp32_vs_fp64(orc):
    P, M, D, W, H, seed = 20000, 1, 0, 200, 72, 14
    s, cams, views = make_scene(P, M, seed, W, H, n_cams=2)
    vp = view_parts(views[1])
    args = (D, M, vp["bg"], W, H, s["loc"], s["sh"], s["opac"], s["scale"], 1.0, s["rot"], vp["view"], vp["proj"], vp["campos"],
            vp["tanx"], vp["tany"])
    r32, r64 = orc.Rasterizer(np.float32), orc.Rasterizer(np.float64)
    r32.forward(*args)
    r64.forward(*args)
    dpix = np.random.default_rng(seed).uniform(-1, 1, (3, H, W)).astype(np.float32)
    g64 = r64.backward(dpix)
    g0 = r32.backward(dpix, flip_margin=0.0)
    assert not g0["flip9"].any()
    n0 = int(_bad(P, g0, g64, g0["abs9"], g0["flip9"]).sum())
    g4 = r32.backward(dpix, flip_margin=1e-4)
    n4 = int(_bad(P, g4, g64, g4["abs9"], g4["flip9"]).sum())
    assert n4 == 0 and n0 > 0, (n0, n4)   # a flipped pair exists in this scene, and the allowance covers exactly that
    assert (g4["flip9"].sum(1) > 0).mean() < 0.05
    # the allowance is monotone in the margin and leaves the gradients themselves untouched
    g3 = r32.backward(dpix, flip_margin=1e-3)
    assert np.all(g3["flip9"] >= g4["flip9"])
    for k in NINE:
        assert np.array_equal(g3[k], g0[k])


def test_forced_flip_is_a_real_alternative_blend(orc):
    """One opaque splat stack on one pixel: flipping the T-stop decision of the entry that is not applied changes the
    front splat's opacity gradient by the closed-form amount, and flip9 reports at least that."""
    W = H = 17
    n = 3
    s = dict(loc=np.array([[0, 0, 1.0 + 0.5 * i] for i in range(n)], np.float32).reshape(-1), scale=np.full(3 * n, 0.2, np.float32),
             rot=np.tile(np.array([1, 0, 0, 0], np.float32), n), opac=np.full(n, 0.95, np.float32), sh=np.zeros(3 * n, np.float32))
    s["sh"][0::3] = [1.0, -0.5, 0.7][:n]
    view = np.eye(4, dtype=np.float32).T.reshape(-1)
    proj = np.zeros(16, np.float32); proj[0] = 1.0; proj[5] = 1.0; proj[10] = 1.0; proj[11] = 1.0
    r = orc.Rasterizer(np.float32)
    r.forward(0, 1, np.zeros(3, np.float32), W, H, s["loc"], s["sh"], s["opac"], s["scale"], 1.0, s["rot"], view, proj,
              np.zeros(3, np.float32), 1.0, 1.0)
    dpix = np.zeros((3, H, W), np.float32); dpix[:, 8, 8] = 1.0
    g = r.backward(dpix, flip_margin=2.0)   # margin > 1: every decision counts as fragile
    assert g["flip9"][:, 8].min() > 0      # every splat's dL_dopacity could move
