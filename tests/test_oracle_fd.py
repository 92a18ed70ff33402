"""Pins the oracle's analytic backward WITHOUT the (absent) upstream source: fp64 central finite
differences of the oracle forward w.r.t. every parameter class.  Scenes avoid the
blend's non-differentiable cut-offs by construction (moderate opacities, big splats, h = 1e-6)."""
import numpy as np
import pytest

import gsplat_amd as gs


def _scene(P, M, seed):
    rng = np.random.default_rng(seed)
    loc = rng.uniform(-1.5, 1.5, (P, 3))
    scale = rng.uniform(0.15, 0.5, (P, 3))
    rot = rng.normal(size=(P, 4))
    rot /= np.linalg.norm(rot, axis=1, keepdims=True)
    rot *= rng.uniform(0.8, 1.2, (P, 1))  # the rasterizer does NOT normalise: gradient is w.r.t. the raw quaternion
    opac = rng.uniform(0.3, 0.9, P)
    sh = rng.uniform(-0.3, 0.3, (P, M, 3))
    sh[:, 0, :] = rng.uniform(-0.5, 1.5, (P, 3))
    return dict(loc=loc, sh=sh, scale=scale, opac=opac, rot=rot)


@pytest.mark.parametrize("M,D", [(1, 0), (4, 1), (9, 2), (16, 3)])
def test_backward_matches_finite_differences(orc, M, D):
    P, W, H = 6, 40, 32  # non-square on purpose: tan_fovx != tan_fovy, focal_x != focal_y
    params = _scene(P, M, 100 + D)
    cam = gs.camera.Camera([3.0, 2.0, -8.0], (0, 0, 0), 50.0)
    vb = gs.camera.view_block(cam, W, H, white=True).astype(np.float64)
    view, proj, campos, tanx, tany = vb[0:16], vb[16:32], vb[32:35], vb[35] * W / H, vb[36]
    bg = np.array([0.3, 0.7, 0.1])
    wpix = np.random.default_rng(1).normal(size=(3, H, W))
    r = orc.Rasterizer(np.float64)

    def f(p):
        out, _ = r.forward(D, M, bg, W, H, p["loc"], p["sh"], p["opac"], p["scale"], 1.0, p["rot"], view, proj, campos, tanx, tany)
        return float((out * wpix).sum())

    f(params)
    g = r.backward(wpix)
    names = dict(loc="dL_dmean3D", sh="dL_dsh", scale="dL_dscale", opac="dL_dopacity", rot="dL_drot")
    rng = np.random.default_rng(2)
    h = 1e-6
    for k, arr in params.items():
        an = g[names[k]].reshape(arr.shape)
        flat_idx = rng.choice(arr.size, size=min(arr.size, 40), replace=False)
        for fi in flat_idx:
            idx = np.unravel_index(fi, arr.shape)
            p = {kk: vv.copy() for kk, vv in params.items()}
            p[k][idx] += h
            lp = f(p)
            p[k][idx] -= 2 * h
            lm = f(p)
            fd = (lp - lm) / (2 * h)
            assert abs(fd - an[idx]) <= 1e-6 * max(1.0, np.abs(an).max()), (k, idx, fd, an[idx])


def test_fp32_oracle_tracks_fp64(orc):
    """The fp32 oracle (the one the GPU is compared with) agrees with its fp64 twin to fp32 accuracy."""
    P, M, D, W, H = 200, 4, 1, 64, 48
    s = gs.synth.random_splats(P, M, 3)
    cams = gs.camera.get_cameras(1)
    vb = gs.camera.train_views(cams, W, H)[0]
    outs = {}
    for dt in (np.float32, np.float64):
        r = orc.Rasterizer(dt)
        out, R = r.forward(D, M, vb[37:40], W, H, s["loc"], s["sh"], s["opac"], s["scale"], 1.0, s["rot"], vb[0:16], vb[16:32],
                           vb[32:35], float(vb[35]), float(vb[36]))
        outs[dt] = (out, R)
    assert outs[np.float32][1] == outs[np.float64][1]
    d = np.abs(outs[np.float32][0] - outs[np.float64][0])
    assert np.quantile(d, 0.999) < 1e-4


@pytest.mark.parametrize("seed", [7, 8, 9])
def test_backward_matches_finite_differences_on_needles(orc, seed):
    """The same pin in the regime the scene sweep lives in (tests/test_gpu_sweep.py): scale axes two orders of magnitude apart, unnormalised
    quaternions, a camera close to the splats.  In fp64 the chain's cancellation costs nothing, so the analytic
    gradient of every parameter class — the scale and rotation of the needles above all — must equal central differences as on the
    well-conditioned scenes; the budget terms the sweep adds for fp32 are about ROUNDING, and this is what says the formulas under them
    are right."""
    P, M, D, W, H = 5, 4, 1, 48, 40
    rng = np.random.default_rng(seed)
    loc = rng.uniform(-0.8, 0.8, (P, 3))
    scale = np.exp(rng.uniform(np.log(0.004), np.log(0.7), (P, 3)))
    scale[0] = [0.9, 0.006, 0.02]                     # ratios of 150 and 45 on one splat
    rot = rng.normal(size=(P, 4))
    rot = rot / np.linalg.norm(rot, axis=1, keepdims=True) * rng.uniform(0.8, 1.2, (P, 1))
    opac = rng.uniform(0.3, 0.9, P)
    sh = rng.uniform(-0.3, 0.3, (P, M, 3))
    sh[:, 0, :] = rng.uniform(-0.5, 1.5, (P, 3))
    params = dict(loc=loc, sh=sh, scale=scale, opac=opac, rot=rot)
    cam = gs.camera.Camera([1.5, 1.0, -4.0], (0, 0, 0), 70.0)
    vb = gs.camera.view_block(cam, W, H, white=True).astype(np.float64)
    view, proj, campos, tanx, tany = vb[0:16], vb[16:32], vb[32:35], vb[35] * W / H, vb[36]
    bg = np.array([0.3, 0.7, 0.1])
    wpix = np.random.default_rng(1).normal(size=(3, H, W))
    r = orc.Rasterizer(np.float64)

    def f(p):
        out, _ = r.forward(D, M, bg, W, H, p["loc"], p["sh"], p["opac"], p["scale"], 1.0, p["rot"], view, proj, campos, tanx, tany)
        return float((out * wpix).sum())

    f(params)
    g = r.backward(wpix)
    names = dict(loc="dL_dmean3D", sh="dL_dsh", scale="dL_dscale", opac="dL_dopacity", rot="dL_drot")
    worst = 0.0
    for k, arr in params.items():
        an = g[names[k]].reshape(arr.shape)
        for fi in range(arr.size) if k != "sh" else range(0, arr.size, 5):
            idx = np.unravel_index(fi, arr.shape)
            h = 1e-7 * max(1.0, abs(arr[idx])) if k != "scale" else 1e-4 * arr[idx]      # a step the 0.006-wide axis can take
            p = {kk: vv.copy() for kk, vv in params.items()}
            p[k][idx] += h
            lp = f(p)
            p[k][idx] -= 2 * h
            lm = f(p)
            fd = (lp - lm) / (2 * h)
            err = abs(fd - an[idx]) / max(1.0, np.abs(an).max())
            worst = max(worst, err)
            assert err <= 2e-5, (k, idx, fd, an[idx])
    print(f"needle scene {seed}: worst |FD - analytic| / max|analytic| = {worst:.2e}")
    assert np.abs(g["dL_dscale"]).max() > 0 and np.abs(g["dL_drot"]).max() > 0
