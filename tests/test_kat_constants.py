"""Known-answer tests for the rasterizer constants that invariance tests cannot see (VERDICT r01 "weak" #2):

  1. near plane:   p_view.z <= 0.2 is culled, the next float above is not          
  2. 1.3*tan_fov:  cov2D uses the CLAMPED t, and the backward masks dL/dt.x / dL/dt.y on the clamped axis and
                   treats the clamped t as independent of t.z                        (A.1, A.8)
  3. T < 1e-4:     the entry that would take T below 1e-4 is NOT applied; n_contrib / final_T / gradients stop there
                                                                                     (A.6, A.7)
  4. 0.99 cap:     alpha = min(0.99, opacity * G); the backward treats the cap as identity (A.7)
  5. det == 0:     a splat whose 2D covariance has an exactly zero determinant is culled (A.1)

Every expected value below is derived in this file (closed forms, or central differences of an independent
one-splat fp64 forward model written here) — not taken from the oracle.  The same cases run on the CPU against the
oracle in fp32 and fp64 (un-marked tests) and on the GPU through the C-ABI rasterizer seam (gpu-marked tests)."""
import math

import numpy as np
import pytest

from util import REC_DTYPE, view_parts  # noqa: F401

SH_C0 = 0.28209479177387814


def simple_camera(tanx=1.0, tany=1.0, bg=(0.0, 0.0, 0.0)):
    """view = identity (+z forward), projview: clip = (x/tanx, y/tany, z, z) -> p_proj = (x/(tanx z), y/(tany z))."""
    view = np.eye(4).T.reshape(-1)
    proj = np.zeros(16)
    proj[0] = 1.0 / tanx; proj[5] = 1.0 / tany; proj[10] = 1.0; proj[11] = 1.0
    return dict(view=view.astype(np.float32), proj=proj.astype(np.float32), campos=np.zeros(3, np.float32), tanx=float(tanx),
                tany=float(tany), bg=np.asarray(bg, np.float32))


def splats(loc, scale, opac, rgb, rot=None):
    """D = 0 splats with the given rgb (sh0 = (rgb - 0.5) / C0)."""
    loc = np.asarray(loc, np.float64).reshape(-1, 3)
    n = loc.shape[0]
    rgb = np.broadcast_to(np.asarray(rgb, np.float64), (n, 3))
    rot = np.tile([1.0, 0, 0, 0], (n, 1)) if rot is None else np.asarray(rot, np.float64).reshape(n, 4)
    return dict(loc=loc.reshape(-1), scale=np.broadcast_to(np.asarray(scale, np.float64), (n, 3)).reshape(-1).copy(),
                opac=np.broadcast_to(np.asarray(opac, np.float64), (n,)).copy(), sh=((rgb - 0.5) / SH_C0).reshape(-1), rot=rot.reshape(-1))


class OracleRun:
    """One forward(+backward) on the CPU oracle in the given precision, with the accessors the cases need."""

    def __init__(self, orc, dtype, s, vp, W, H, mod=1.0):
        self.r = orc.Rasterizer(dtype)
        self.dtype = np.dtype(dtype)
        self.P = s["opac"].size
        self.W, self.H = W, H
        self.out, self.R = self.r.forward(0, 1, vp["bg"], W, H, s["loc"], s["sh"], s["opac"], s["scale"], mod, s["rot"], vp["view"],
                                          vp["proj"], vp["campos"], vp["tanx"], vp["tany"])
        self._f32 = self.r if self.dtype == np.float32 else None

    def radii(self):
        if self._f32 is None:   # orc_get exposes the fp32 state only: a culled splat shows as R == 0 / background
            return None
        return self.r.get("radii")

    def conic(self):
        return None if self._f32 is None else self.r.get("conic_opacity").reshape(self.P, 4)[:, :3]

    def n_contrib(self):
        return None if self._f32 is None else self.r.get("n_contrib").reshape(self.H, self.W)

    def final_T(self):
        return None if self._f32 is None else self.r.get("final_T").reshape(self.H, self.W)

    def backward(self, dpix):
        return self.r.backward(dpix)


class SeamRun:
    """The same through gs_rasterize_forward / gs_rasterize_backward on the GPU."""

    def __init__(self, s, vp, W, H, mod=1.0):
        from util import SeamRaster
        self.sr = SeamRaster()
        self.P, self.W, self.H = s["opac"].size, W, H
        s32 = {k: np.asarray(v, np.float32) for k, v in s.items()}
        self.out, self.R = self.sr.forward(s32, 0, 1, vp, W, H, mod)
        self.dtype = np.dtype(np.float32)

    def _rec(self):
        return self.sr.field("geometry", "record", np.uint8).view(REC_DTYPE)

    def radii(self):
        tiles = self.sr.field("geometry", "tiles_touched", np.uint32)
        return np.where(tiles > 0, self._rec()["radius"], 0)

    def conic(self):
        rec = self._rec()
        return np.stack([rec["conA"], rec["conB"], rec["conC"]], 1)

    def n_contrib(self):
        return self.sr.field("image", "n_contrib", np.uint32).reshape(self.H, self.W)

    def final_T(self):
        return self.sr.field("image", "final_T", np.float32).reshape(self.H, self.W)

    def backward(self, dpix):
        return self.sr.backward(np.asarray(dpix, np.float32))


def runners(orc):
    return [("oracle-f32", lambda *a, **k: OracleRun(orc, np.float32, *a, **k)), ("oracle-f64", lambda *a, **k: OracleRun(orc, np.float64, *a, **k))]


def gpu_runner():
    return [("hip", lambda *a, **k: SeamRun(*a, **k))]


# ---------------------------------------------------------------------------------------------------------
# 1. near plane
# ---------------------------------------------------------------------------------------------------------
def case_near_plane(make, dtype):
    W = H = 33
    vp = simple_camera()
    ft = np.float32 if np.dtype(dtype) == np.float32 else np.float64
    z_on = ft(0.2)                              # exactly the constant the implementation compares with: culled (<=)
    z_above = np.nextafter(z_on, ft(1.0))       # the next representable depth: visible
    for z, visible in [(z_on, False), (z_above, True), (ft(0.19), False), (ft(0.25), True)]:
        run = make(splats([0, 0, float(z)], 0.01, 0.8, [1.0, 0.5, 0.25]), vp, W, H)
        assert (run.R > 0) == visible, (float(z), run.R)
        rad = run.radii()
        if rad is not None:
            assert (rad[0] > 0) == visible
        centre = run.out[:, 16, 16]
        if visible:
            assert centre[0] > 0.5       # the splat covers the centre pixel (s_px = 16*0.01/0.2 = 0.8 px, +0.3 low-pass)
        else:
            assert np.all(run.out == 0.0)


def test_near_plane_cpu(orc):
    for name, make in runners(orc):
        case_near_plane(make, np.float64 if name.endswith("f64") else np.float32)


@pytest.mark.gpu
def test_near_plane_gpu():
    case_near_plane(gpu_runner()[0][1], np.float32)


# ---------------------------------------------------------------------------------------------------------
# 2. the 1.3 * tan_fov clamp, forward and backward
# ---------------------------------------------------------------------------------------------------------
def _one_splat_model(W, H, tanx, tany, sig, opac, rgb, bg):
    """Independent fp64 forward of ONE isotropic splat under view = identity, as a function of the quantities the
    upstream backward differentiates separately: the (clamped) t.x, t.y used in the Jacobian, t.z, and the projected
    pixel centre.  Returns (image(3,H,W), conic)."""
    fx, fy = W / (2 * tanx), H / (2 * tany)
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)

    def model(txc, tyc, tz, mx, my):
        J = np.array([[fx / tz, 0.0, -fx * txc / tz ** 2], [0.0, fy / tz, -fy * tyc / tz ** 2]])
        cov = (sig ** 2) * (J @ J.T) + 0.3 * np.eye(2)
        a, b, c = cov[0, 0], cov[0, 1], cov[1, 1]
        det = a * c - b * b
        con = np.array([c / det, -b / det, a / det])
        dx, dy = mx - xs, my - ys
        power = -0.5 * (con[0] * dx * dx + con[2] * dy * dy) - con[1] * dx * dy
        alpha = np.minimum(0.99, opac * np.exp(power))
        alpha = np.where((power > 0) | (alpha < 1.0 / 255.0), 0.0, alpha)
        img = np.asarray(rgb)[:, None, None] * alpha + (1.0 - alpha) * np.asarray(bg)[:, None, None]
        return img, con, alpha
    return model, fx, fy


def case_fov_clamp(make, dtype, tol):
    W, H, tanx, tany = 64, 48, 0.6, 0.45
    sig, opac, rgb, bg = 0.9, 0.6, np.array([0.9, 0.3, 0.6]), np.array([0.1, 0.2, 0.3])
    model, fx, fy = _one_splat_model(W, H, tanx, tany, sig, opac, rgb, bg)
    tz = 4.0
    for xr, yr in [(1.45, 0.2), (0.3, -1.5), (0.5, 0.4)]:     # clamped in x, clamped in y (negative side), not clamped
        mean = np.array([xr * tanx * tz, yr * tany * tz, tz])
        limx, limy = 1.3 * tanx, 1.3 * tany
        txc = min(limx, max(-limx, mean[0] / tz)) * tz
        tyc = min(limy, max(-limy, mean[1] / tz)) * tz
        mask = np.array([abs(xr) <= 1.3, abs(yr) <= 1.3], float)
        pix = lambda m: (((m[0] / (tanx * (m[2] + 1e-7)) + 1) * W - 1) * 0.5, ((m[1] / (tany * (m[2] + 1e-7)) + 1) * H - 1) * 0.5)
        mx, my = pix(mean)
        img, con, alpha = model(txc, tyc, tz, mx, my)
        run = make(splats(mean, sig, opac, rgb), simple_camera(tanx, tany, bg), W, H)
        # forward: the conic is the one of the CLAMPED Jacobian (the unclamped one differs by far more than the tolerance)
        got = run.conic()
        if got is not None:
            assert np.allclose(got[0], con, rtol=5e-5, atol=0), (got[0], con)
            if mask.min() == 0:
                _, con_unclamped, _ = model(mean[0], mean[1], tz, mx, my)
                assert not np.allclose(got[0], con_unclamped, rtol=1e-2, atol=0)
        assert np.abs(run.out - img).max() < 2e-4
        assert (alpha > 0.05).sum() > 20      # the splat reaches into the image although its centre may lie outside
        # backward.  L = sum(dpix * image) with dpix supported well inside the footprint (no threshold pixel moves).
        rng = np.random.default_rng(3)
        dpix = rng.uniform(-1, 1, (3, H, W)) * (alpha > 0.05)
        L = lambda *a: float((model(*a)[0] * dpix).sum())
        h = 1e-5
        dtxc = (L(txc + h, tyc, tz, mx, my) - L(txc - h, tyc, tz, mx, my)) / (2 * h)
        dtyc = (L(txc, tyc + h, tz, mx, my) - L(txc, tyc - h, tz, mx, my)) / (2 * h)
        dtz = (L(txc, tyc, tz + h, mx, my) - L(txc, tyc, tz - h, mx, my)) / (2 * h)     # clamped t held fixed, as upstream does
        dmx = (L(txc, tyc, tz, mx + h, my) - L(txc, tyc, tz, mx - h, my)) / (2 * h)
        dmy = (L(txc, tyc, tz, mx, my + h) - L(txc, tyc, tz, mx, my - h)) / (2 * h)
        # projection path: exact derivative of the pixel centre w.r.t. the mean
        zz = tz + 1e-7
        jac_pix = np.array([[0.5 * W / (tanx * zz), 0.0, -0.5 * W * mean[0] / (tanx * zz * zz)],
                            [0.0, 0.5 * H / (tany * zz), -0.5 * H * mean[1] / (tany * zz * zz)]])
        want = np.array([mask[0] * dtxc, mask[1] * dtyc, dtz]) + jac_pix.T @ np.array([dmx, dmy])
        g = run.backward(dpix)
        got_g = np.asarray(g["dL_dmean3D"], np.float64)[:3]
        scale = np.abs(want).max()
        assert np.all(np.abs(got_g - want) <= tol * scale), (xr, yr, got_g, want)
        # the mask matters in the clamped cases: without it the expectation would be off by much more than the tolerance
        if mask.min() == 0:
            unmasked = np.array([dtxc, dtyc, dtz]) + jac_pix.T @ np.array([dmx, dmy])
            assert np.abs(unmasked - want).max() > 50 * tol * scale


def test_fov_clamp_cpu(orc):
    for name, make in runners(orc):
        case_fov_clamp(make, None, 2e-6 if name.endswith("f64") else 3e-4)


@pytest.mark.gpu
def test_fov_clamp_gpu():
    case_fov_clamp(gpu_runner()[0][1], None, 3e-4)


# ---------------------------------------------------------------------------------------------------------
# 3. T < 1e-4 stops the blend BEFORE applying the entry;  4. the 0.99 cap is an identity in the backward
# ---------------------------------------------------------------------------------------------------------
def _stack(n, opac, W=17):
    """n isotropic splats on the optical axis, front to back, all centred exactly on pixel ((W-1)/2, (W-1)/2)."""
    rgb = np.array([[0.9, 0.2, 0.4], [0.1, 0.8, 0.3], [0.5, 0.5, 0.9], [0.7, 0.1, 0.6], [0.3, 0.9, 0.2]])[:n]
    loc = [[0, 0, 2.0 + 0.25 * i] for i in range(n)]
    return splats(loc, 0.3, opac, rgb), rgb


def case_t_stop(make, rtol):
    W = H = 17
    c = 8
    bg = np.array([0.25, 0.5, 0.75])
    a = 0.95
    s, rgb = _stack(5, a)
    run = make(s, simple_camera(1.0, 1.0, bg), W, H)
    # T after k entries = 0.05^k: 0.05, 2.5e-3, 1.25e-4, then 6.25e-6 < 1e-4 -> the 4th entry is not applied
    T = [0.05 ** k for k in range(4)]
    want = sum(rgb[k] * a * T[k] for k in range(3)) + T[3] * bg
    assert np.allclose(run.out[:, c, c], want, rtol=0, atol=2e-6), (run.out[:, c, c], want)
    with_fourth = want - T[3] * bg + rgb[3] * a * T[3] + 0.05 ** 4 * bg
    assert np.abs(with_fourth - want).max() > 1e-5          # the two conventions are distinguishable at this tolerance
    if run.n_contrib() is not None:
        assert run.n_contrib()[c, c] == 3
        assert abs(run.final_T()[c, c] - T[3]) < 1e-9
    dp = np.array([0.7, -0.4, 0.9])
    dpix = np.zeros((3, H, W)); dpix[:, c, c] = dp
    g = run.backward(dpix)
    # dL/dalpha_j = sum_c dp_c * ( T_j * (c_j - behind_j) ) - T_final / (1 - alpha) * (bg . dp),  behind_j = colour accumulated behind j
    behind = np.zeros(3)
    want_op = np.zeros(5)
    for j in (2, 1, 0):
        want_op[j] = float(dp @ ((rgb[j] - behind) * T[j])) - T[3] / (1 - a) * float(bg @ dp)
        behind = a * rgb[j] + (1 - a) * behind
    got = np.asarray(g["dL_dopacity"], np.float64)
    assert np.allclose(got[:3], want_op[:3], rtol=rtol, atol=1e-9), (got, want_op)
    assert got[3] == 0.0 and got[4] == 0.0                  # entries at and behind the stop get no gradient at all
    assert not np.asarray(g["dL_dmean3D"]).reshape(5, 3)[3:].any()
    col = np.asarray(g["dL_dcolor"], np.float64).reshape(5, 3)
    for j in range(3):
        assert np.allclose(col[j], a * T[j] * dp, rtol=rtol, atol=1e-12)
    assert not col[3:].any()


def case_alpha_cap(make, rtol):
    W = H = 17
    c = 8
    bg = np.array([0.25, 0.5, 0.75])
    s, rgb = _stack(1, 1.0)                                   # opacity 1 -> alpha = min(0.99, 1 * exp(0)) = 0.99 at the centre
    run = make(s, simple_camera(1.0, 1.0, bg), W, H)
    assert np.allclose(run.out[:, c, c], 0.99 * rgb[0] + 0.01 * bg, rtol=0, atol=2e-6)
    dp = np.array([0.7, -0.4, 0.9])
    dpix = np.zeros((3, H, W)); dpix[:, c, c] = dp
    g = run.backward(dpix)
    # upstream differentiates through the cap as if it were not there: dL/dopacity = G * dL/dalpha with G = 1,
    # dL/dalpha = dp . (c - bg * T_final / (1 - alpha)) = dp . (c - bg); the true derivative of min(0.99, .) would be 0
    want = float(dp @ (rgb[0] - bg))
    got = float(np.asarray(g["dL_dopacity"])[0])
    assert abs(want) > 0.1 and abs(got - want) <= rtol * abs(want) + 1e-9, (got, want)


def test_t_stop_and_alpha_cap_cpu(orc):
    for name, make in runners(orc):
        rt = 1e-9 if name.endswith("f64") else 2e-5
        case_t_stop(make, max(rt, 1e-7))
        case_alpha_cap(make, max(rt, 1e-7))


@pytest.mark.gpu
def test_t_stop_and_alpha_cap_gpu():
    make = gpu_runner()[0][1]
    case_t_stop(make, 2e-5)
    case_alpha_cap(make, 2e-5)


# ---------------------------------------------------------------------------------------------------------
# 5. det == 0
# ---------------------------------------------------------------------------------------------------------
def case_det_zero(make, dtype):
    """A needle (scale (L, 0, 0)) under the unnormalised quaternion (0.5, 0, 0, 0.5) has the rotation matrix columns
    (0.5, 0.5, 0), so its 3D covariance is L^2 [[.25, .25, 0], [.25, .25, 0], [0, 0, 0]] exactly.  On the optical axis with
    fx = fy = 32, t.z = 1 the 2D covariance is a = b = c = 256 L^2 (+0.3 on a, c).  With L chosen so that 256 L^2 is a
    power of two large enough to absorb the 0.3, a * c - b * b is EXACTLY zero in that precision: upstream returns
    before writing anything for the splat."""
    W = H = 64
    bg = np.array([0.2, 0.4, 0.6])
    f64 = np.dtype(dtype) == np.float64
    L = 2.0 ** 26 if f64 else 512.0                         # 256 L^2 = 2^60 (fp64) / 2^26 (fp32): ulp 256 / 8 >> 0.3
    q = [0.5, 0.0, 0.0, 0.5]
    s = splats([0, 0, 1.0], [L, 0.0, 0.0], 0.9, [0.9, 0.1, 0.1], rot=q)
    run = make(s, simple_camera(1.0, 1.0, bg), W, H)
    assert run.R == 0
    if run.radii() is not None:
        assert run.radii()[0] == 0
    assert np.allclose(run.out, bg[:, None, None], rtol=0, atol=1e-7)
    g = run.backward(np.ones((3, H, W)))
    for k in ("dL_dmean3D", "dL_dscale", "dL_drot", "dL_dopacity", "dL_dsh"):
        assert not np.asarray(g[k]).any(), k
    # control: a slightly shorter needle (0.3 is no longer absorbed exactly) is NOT culled
    s2 = splats([0, 0, 1.0], [0.05, 0.0, 0.0], 0.9, [0.9, 0.1, 0.1], rot=q)
    run2 = make(s2, simple_camera(1.0, 1.0, bg), W, H)
    assert run2.R > 0


def test_det_zero_cpu(orc):
    for name, make in runners(orc):
        case_det_zero(make, np.float64 if name.endswith("f64") else np.float32)


@pytest.mark.gpu
def test_det_zero_gpu():
    case_det_zero(gpu_runner()[0][1], np.float32)
