"""CPU: the decision-flip allowance of the oracle's backward (gs_oracle.cpp, render_backward; used by the GPU parity
tests) is checked here against an independent "other implementation": the same oracle in fp64.  fp32 and fp64 take a
different branch at a few (pixel, splat) pairs that sit on a blend threshold; the nine pixel-stage sums of the two must
agree within 1e-4 * sum|term| + flip9(margin) for EVERY splat once the margin covers fp32 rounding, and the allowance
must be what closes the gap (in the dense scene some splats are out of budget without it)."""
import numpy as np

import gsplat_amd as gs
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


def _blend_pixel(r, W, px, py, bg, flip_at=None):
    """One pixel's forward blend in plain fp32 numpy scalars, written here independently of the
    oracle's C++; flip_at = (list position, kind): that decision (1: alpha test, 2: T test) is taken the other way.
    Returns (colour incl. background, T, last contributor, [(position, kind, relative distance from the threshold)])."""
    f = np.float32
    gx = (W + 15) // 16
    beg, end = r.get("ranges").reshape(-1, 2)[(py // 16) * gx + px // 16]
    pl, m2, co, rgb = r.get("point_list"), r.get("means2D").reshape(-1, 2), r.get("conic_opacity").reshape(-1, 4), r.get("rgb").reshape(-1, 3)
    T, C, last, dist = f(1.0), np.zeros(3, f), 0, []
    for n, k in enumerate(range(beg, end), 1):
        i = pl[k]
        dx, dy = m2[i, 0] - f(px), m2[i, 1] - f(py)
        power = f(-0.5) * (co[i, 0] * dx * dx + co[i, 2] * dy * dy) - co[i, 1] * dx * dy
        if power > 0:
            continue
        alpha = min(f(0.99), co[i, 3] * np.exp(power, dtype=f))
        skip = alpha < f(1.0) / f(255.0)
        dist.append((k, 1, abs(float(alpha) - 1.0 / 255.0) * 255.0))
        if flip_at == (k, 1):
            skip = not skip
        if skip:
            continue
        test_T = T * (f(1.0) - alpha)
        stop = test_T < f(0.0001)
        dist.append((k, 2, abs(float(test_T) - 0.0001) * 10000.0))
        if flip_at == (k, 2):
            stop = not stop
        if stop:
            break
        C = C + rgb[i] * alpha * T
        T, last = test_T, n
    return (C + T * np.asarray(bg, f)).astype(f), T, last, dist


def test_pixel_check_accepts_admissible_blends_and_nothing_else(orc):
    """orc.check_pixels (the forward counterpart of the flip allowance; tests/test_gpu_raster.py::_check_forward uses it to
    compare EVERY pixel instead of excluding the fragile ones): the fp32 oracle against itself is the nominal blend
    everywhere; "another implementation" that takes the other branch at the scene's most fragile decision — blended here by
    an independent numpy restatement of the pixel loop — is admissible at that pixel exactly when the margin reaches that
    decision, and refused otherwise; results that are no blend of the pixel are refused."""
    P, M, D, W, H, seed = 20000, 1, 0, 200, 72, 14
    s, cams, views = make_scene(P, M, seed, W, H, n_cams=2)
    vp = view_parts(views[1])
    r32 = orc.Rasterizer(np.float32)
    out32, _ = r32.forward(D, M, vp["bg"], W, H, s["loc"], s["sh"], s["opac"], s["scale"], 1.0, s["rot"], vp["view"], vp["proj"], vp["campos"],
                           vp["tanx"], vp["tany"])
    fT, nc, margin = r32.get("final_T"), r32.get("n_contrib"), r32.get("margin")
    st, leaves = orc.check_pixels(r32, out32, fT, nc)
    assert not st.any() and np.all(leaves == 1)                 # the nominal blend is examined first and is the one
    # the independent pixel blender agrees with the oracle on the nominal blend of the most fragile pixel ...
    pix = int(np.argmin(margin))
    py, px = divmod(pix, W)
    col, T, last, dist = _blend_pixel(r32, W, px, py, vp["bg"])
    assert last == nc[pix] and abs(float(T) - fT[pix]) <= 1e-6 * fT[pix] + 1e-12 and np.allclose(col, out32[:, py, px], rtol=1e-5, atol=1e-6)
    k, kind, d = min(dist, key=lambda t: t[2])
    assert abs(d - margin[pix]) <= 0.5 * margin[pix] + 1e-6 and d < 1e-3    # (fp64 here, fp32 there)
    # ... and its blend with THAT decision inverted is another admissible blend of the pixel — if the margin reaches it
    col_f, T_f, last_f, _ = _blend_pixel(r32, W, px, py, vp["bg"], flip_at=(k, kind))
    assert last_f != last or abs(float(T_f) - float(T)) > 1e-3 * float(T) or np.abs(col_f - col).max() > 1e-4
    foreign, fT2, nc2 = out32.copy(), fT.copy(), nc.copy()
    foreign[:, py, px], fT2[pix], nc2[pix] = col_f, T_f, last_f
    st_in, leaves_in = orc.check_pixels(r32, foreign, fT2, nc2, alpha_margin=3.0 * d + 1e-6, T_margin=3.0 * d + 1e-6)
    assert st_in[pix] == 1 and leaves_in[pix] >= 2 and (st_in != 0).sum() == 1
    st_out, _ = orc.check_pixels(r32, foreign, fT2, nc2, alpha_margin=0.0, T_margin=0.0)
    assert st_out[pix] == 2 and (st_out != 0).sum() == 1
    # a pixel whose colour is off by more than any admissible blend allows is refused, with or without fragile decisions
    bad = out32.copy()
    bad[:, 5, 7] += 0.05
    st2, _ = orc.check_pixels(r32, bad, fT, nc)
    assert st2[5 * W + 7] == 2 and (st2 != 0).sum() == 1
    nc3 = nc.copy(); nc3[11] += 1
    st3, _ = orc.check_pixels(r32, out32, fT, nc3)
    assert st3[11] == 2 and (st3 != 0).sum() == 1
    # a wide margin makes many decisions fragile: the trees grow, the verdict on the oracle's own result does not change
    st4, leaves4 = orc.check_pixels(r32, out32, fT, nc, alpha_margin=0.05, T_margin=0.05)
    assert not st4.any()


def test_pixel_check_exponent_conditioning(orc):
    """exp_cond of orc.check_pixels: one big needle splat seen far along its long axis — the three products of its exponent cancel by
    three orders of magnitude — blended over a black background by an independent numpy pixel loop that evaluates the SAME exponent
    in a different fp32 order (Horner form with the conic pre-scaled, what a GPU kernel would do).  The two fp32 orders differ by more
    than 1e-4 relative in alpha at some pixels: refused without the term, admissible with exp_cond = 4, and an error eight times
    the conditioning bound is refused with it."""
    f = np.float32
    W = H = 48
    cam = gs.camera.Camera((0.0, 0.0, 10.0), (0.0, 0.0, 0.0), 60.0)
    views = gs.camera.train_views([cam], W, H)
    vp = view_parts(views[1])     # the black pass
    s = dict(loc=np.array([9.0, 6.0, 0.0], f), scale=np.array([6.0, 0.05, 0.05], f), rot=np.array([0.9239, 0.0, 0.0, 0.3827], f),
             opac=np.array([0.9], f), sh=np.array([1.0, 0.5, 0.2], f))
    r = orc.Rasterizer(f)
    out, R = r.forward(0, 1, vp["bg"], W, H, s["loc"], s["sh"], s["opac"], s["scale"], 1.0, s["rot"], vp["view"], vp["proj"], vp["campos"], vp["tanx"], vp["tany"])
    assert R > 0
    m2, co, rgb = r.get("means2D").reshape(-1, 2)[0], r.get("conic_opacity").reshape(-1, 4)[0], r.get("rgb").reshape(-1, 3)[0]
    fT, last = r.get("final_T"), r.get("n_contrib")
    ys, xs = np.mgrid[0:H, 0:W]
    dx, dy = (m2[0] - xs.astype(f)).astype(f), (m2[1] - ys.astype(f)).astype(f)
    power = (f(-0.5) * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy).astype(f)            # the reference's order
    a2, b2, c2 = f(-0.5) * co[0], -co[1], f(-0.5) * co[2]
    other = (dy * (dy * c2) + dx * (dy * b2 + dx * a2)).astype(f)                                     # another fp32 order of the same polynomial
    mag = 0.5 * (abs(co[0]) * dx * dx + abs(co[2]) * dy * dy) + np.abs(co[1] * dx * dy)
    blended = (last.reshape(H, W) == 1)
    assert blended.sum() > 50 and (mag[blended] / np.abs(power[blended])).max() > 300       # the cancellation this test is about
    def image(pw):
        alpha = np.minimum(f(0.99), co[3] * np.exp(pw, dtype=f)).astype(f)
        col = np.zeros((3, H, W), f)
        T = np.ones((H, W), f)
        for c in range(3):
            col[c][blended] = (rgb[c] * alpha * f(1.0))[blended]
        T[blended] = (f(1.0) - alpha)[blended]
        return col, T
    col, T = image(other)
    rel = np.abs(col[0] - out.reshape(3, H, W)[0])[blended] / np.maximum(out.reshape(3, H, W)[0][blended], 1e-3)
    assert rel.max() > 1e-4, rel.max()          # the two orders really are further apart than the plain bar
    st0, _ = orc.check_pixels(r, col, T, last, alpha_margin=0.0, T_margin=0.0)
    st4, _ = orc.check_pixels(r, col, T, last, alpha_margin=0.0, T_margin=0.0, exp_cond=4.0)
    assert (st0 == 2).sum() > 0 and (st4 != 0).sum() == 0, ((st0 == 2).sum(), (st4 != 0).sum())
    # an error of 32 x 2^-24 x magnitude in the exponent is outside what exp_cond = 4 admits
    col, T = image((power.astype(np.float64) + 32.0 * 2.0 ** -24 * mag).astype(f))
    st, _ = orc.check_pixels(r, col, T, last, alpha_margin=0.0, T_margin=0.0, exp_cond=4.0)
    assert (st == 2).sum() > 0


def test_power_sign_decision_on_the_ridge_of_a_needle(orc):
    """The third discrete decision of the blend, `power > 0: skip the pair`.  On the ridge line of a needle splat thousands of pixels long
    the quadratic form is ~1e-5 while its three products are ~100 (it takes a needle this long: the 0.3 px the projection adds to every
    axis bounds the ratio by sigma_long^2 / 0.3): the SIGN of the computed power there belongs to the fp32 evaluation order.  With
    exp_cond, a pair whose |power| is below that many units of its conditioning (2^-24 m) may be taken either way: an image in which
    the most fragile such pair of the scene is decided the other way is admissible at that pixel (status 1) exactly when exp_cond
    reaches its distance, and refused (status 2) below it; the backward's flip allowance covers the same pair (power_ulps)."""
    f = np.float32
    W = H = 65                                                    # odd: the origin projects onto the centre of pixel (32, 32)
    cam = gs.camera.Camera((0.0, 0.0, 10.0), (0.0, 0.0, 0.0), 60.0)
    vp = view_parts(gs.camera.train_views([cam], W, H)[1])     # the black pass
    # a needle through the origin along the image diagonal: the pixels (32 + k, 32 +- k) sit ON its ridge
    s = dict(loc=np.zeros(3, f), scale=np.array([300.0, 0.02, 0.02], f), rot=np.array([0.9238795, 0.0, 0.0, 0.3826834], f),
             opac=np.array([0.8], f), sh=np.array([1.2, 0.4, 0.9], f))
    r = orc.Rasterizer(f)
    out, R = r.forward(0, 1, vp["bg"], W, H, s["loc"], s["sh"], s["opac"], s["scale"], 1.0, s["rot"], vp["view"], vp["proj"], vp["campos"], vp["tanx"], vp["tany"])
    assert R > 0
    m2, co, rgb = r.get("means2D").reshape(-1, 2)[0], r.get("conic_opacity").reshape(-1, 4)[0], r.get("rgb").reshape(-1, 3)[0]
    ys, xs = np.mgrid[0:H, 0:W]
    dx, dy = (m2[0] - xs.astype(f)).astype(f), (m2[1] - ys.astype(f)).astype(f)
    power = (f(-0.5) * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy).astype(f)
    mag = 0.5 * (abs(float(co[0])) * dx.astype(np.float64) ** 2 + abs(float(co[2])) * dy.astype(np.float64) ** 2) + np.abs(float(co[1]) * dx.astype(np.float64) * dy)
    ulps = np.where(mag > 0, np.abs(power.astype(np.float64)) / (2.0 ** -24 * mag + 1e-300), np.inf)     # (not the centre pixel: nothing cancels there)
    fT, last = r.get("final_T").reshape(H, W), r.get("n_contrib").reshape(H, W)
    py, px = np.unravel_index(np.argmin(ulps), ulps.shape)
    d = float(ulps[py, px])
    assert d < 64.0, d                       # the scene does hold a pair whose power is a few ulps of its conditioning from zero
    skipped = bool(power[py, px] > 0)
    assert skipped == (last[py, px] == 0)
    # the same image with THAT decision inverted: blended where the oracle skipped, or the other way round
    col, T, lst = out.reshape(3, H, W).copy(), fT.copy(), last.copy()
    if skipped:
        alpha = min(f(0.99), co[3] * np.exp(power[py, px], dtype=f))
        col[:, py, px] = rgb * alpha
        T[py, px], lst[py, px] = f(1.0) - alpha, 1
    else:
        col[:, py, px], T[py, px], lst[py, px] = 0.0, 1.0, 0
    below, _ = orc.check_pixels(r, col, T, lst, alpha_margin=0.0, T_margin=0.0, exp_cond=d * 0.5)
    above, leaves = orc.check_pixels(r, col, T, lst, alpha_margin=0.0, T_margin=0.0, exp_cond=d * 2.0 + 1.0)
    pix = py * W + px
    assert below[pix] == 2 and above[pix] == 1 and leaves[pix] >= 2
    assert (np.delete(above, pix) == 0).all()
    # the backward: without power_ulps no allowance at a pixel whose only fragile decision is this one, with it there is
    dpix = np.zeros((3, H, W), f)
    dpix[:, py, px] = 1.0
    g0 = r.backward(dpix.reshape(-1), want_abs=True, flip_margin=1e-6, want_cond=True, power_ulps=0.0)
    g1 = r.backward(dpix.reshape(-1), want_abs=True, flip_margin=1e-6, want_cond=True, power_ulps=d * 2.0 + 1.0)
    assert g0["flip9"].sum() == 0 and g1["flip9"][0, :3].min() > 0


def test_frame_check_accepts_quantised_admissible_blends_only(orc):
    """orc.check_frame (orc_check_frame_f32), the byte-level sibling of check_pixels used on Trainer::render's RGBA8 output: the oracle's own
    image quantised with imageFloatToInt is the nominal blend everywhere; a byte moved one step is accepted only where the float sits within the
    pixel tolerance of the k / 256 boundary between the two values; a byte moved two steps, or a wrong alpha byte, never."""
    import util
    P, M, D, W, H = 900, 4, 1, 96, 80
    s, cams, views = util.make_scene(P, M, 31, W, H)
    vp = util.view_parts(views[0])
    r, img, _ = util.oracle_forward(orc, s, D, M, vp, W, H)
    frame = orc.image_float_to_int(img, W, H)
    status, _ = orc.check_frame(r, frame)
    assert not status.any()
    v = img.reshape(3, -1).astype(np.float64)
    # pixels whose red channel lies far from a byte boundary (> 20 x the tolerance) and is not clamped
    frac = v[0] * 256.0 - np.floor(v[0] * 256.0)
    far = np.flatnonzero((frac > 0.3) & (frac < 0.7) & (v[0] > 0.05) & (v[0] < 0.9))[:50]
    assert far.size >= 10
    moved = frame.copy()
    moved[far] += 1                                    # red + 1
    status, _ = orc.check_frame(r, moved)
    assert (status[far] == 2).all() and int((status >= 2).sum()) == far.size
    # a float ON a boundary (within tolerance): both neighbouring bytes are accepted
    near = np.flatnonzero((np.minimum(frac, 1.0 - frac) / 256.0 < 0.5e-4 * np.maximum(v[0], 1e-3)) & (v[0] > 0.05) & (v[0] < 0.9))
    if near.size:
        other = frame.copy()
        up = frac[near] < 0.5                          # just above the boundary: the byte below is the alternative
        other[near] = np.where(up, frame[near] - 1, frame[near] + 1)
        status, _ = orc.check_frame(r, other)
        assert not (status[near] >= 2).any()
    two = frame.copy(); two[far[:5]] += 2
    assert (orc.check_frame(r, two)[0][far[:5]] == 2).all()
    alpha = frame.copy(); alpha[far[0]] &= 0x00FFFFFF
    assert orc.check_frame(r, alpha)[0][far[0]] == 2
