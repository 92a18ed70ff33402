"""Known-answer cases, structural invariants and the reference's own four kernels for the oracle
(SURVEY §4.3, §8c(ii)-(iii)).  These are the pins the oracle has in place of reference fixtures."""
import math

import numpy as np

import gsplat_amd as gs


def _axis_camera(W, H, fov=60.0, dist=10.0):
    cam = gs.camera.Camera([0.0, 0.0, -dist], (0, 0, 0), fov)
    return cam, gs.camera.view_block(cam, W, H, white=False)


def test_single_isotropic_splat_closed_form(orc):
    W = H = 65
    cam, vb = _axis_camera(W, H)
    sigma, opacity = 0.2, 0.6
    sh0 = np.array([1.0, -0.5, 0.25], np.float32)
    r = orc.Rasterizer(np.float32)
    bg = np.array([0.2, 0.4, 0.9], np.float32)
    out, R = r.forward(0, 1, bg, W, H, [0, 0, 0], sh0, [opacity], [sigma] * 3, 1.0, [1, 0, 0, 0], vb[0:16], vb[16:32],
                       vb[32:35], float(vb[35]), float(vb[36]))
    focal = W / (2 * math.tan(math.radians(30)))
    s_px2 = (focal * sigma / 10.0) ** 2
    conic = r.get("conic_opacity")
    assert abs(conic[0] - 1 / (s_px2 + 0.3)) < 1e-5 * conic[0] and abs(conic[1]) < 1e-7 and abs(conic[2] - conic[0]) < 1e-6
    # isotropic: mid^2 - det = 0, so the max(0.1, .) floor adds sqrt(0.1) to the larger eigenvalue
    assert r.get("radii")[0] == math.ceil(3 * math.sqrt(s_px2 + 0.3 + math.sqrt(0.1)))
    m2 = r.get("means2D")
    assert abs(m2[0] - 32.0) < 1e-3 and abs(m2[1] - 32.0) < 1e-3  # ndc2Pix of the optical axis = (W-1)/2
    col = np.maximum(0.28209479177387814 * sh0 + 0.5, 0)
    centre = out[:, 32, 32]
    assert np.allclose(centre, opacity * col + (1 - opacity) * bg, atol=2e-5)
    assert r.get("n_contrib").reshape(H, W)[32, 32] == 1
    assert np.allclose(out[:, 0, 0], bg)  # far corner: alpha < 1/255 -> pure background


def test_zero_sh_is_exactly_half_grey(orc):
    """initFieldGrid starts with all SH = 0 -> colour exactly 0.5 (src/ui/UiFrame.cpp:143-151)."""
    W = H = 48
    cam, vb = _axis_camera(W, H)
    r = orc.Rasterizer(np.float32)
    r.forward(1, 4, [0, 0, 0], W, H, [0.3, -0.2, 0.1], np.zeros(12), [1.0], [0.05] * 3, 1.0, [0, 0, 0, 1], vb[0:16], vb[16:32],
              vb[32:35], float(vb[35]), float(vb[36]))
    assert np.all(r.get("rgb")[:3] == 0.5)


def test_white_minus_black_background_is_final_T(orc):
    P, M, W, H = 400, 4, 96, 64
    s = gs.synth.random_splats(P, M, 17)
    views = gs.camera.train_views(gs.camera.get_cameras(1), W, H)
    outs = []
    for v in (0, 1):
        vb = views[v]
        r = orc.Rasterizer(np.float32)
        out, _ = r.forward(1, M, vb[37:40], W, H, s["loc"], s["sh"], s["opac"], s["scale"], 1.0, s["rot"], vb[0:16], vb[16:32],
                           vb[32:35], float(vb[35]), float(vb[36]))
        outs.append(out)
    fT = r.get("final_T").reshape(H, W)
    for c in range(3):
        assert np.allclose(outs[0][c] - outs[1][c], fT, atol=1e-6)


def test_structural_invariants(orc):
    P, M, W, H = 1000, 4, 256, 256  # BASELINE cfg1 shape
    s = gs.synth.random_splats(P, M, gs.synth.seed_for(1))
    vb = gs.camera.train_views(gs.camera.get_cameras(1), W, H)[0]
    r = orc.Rasterizer(np.float32)
    out, R = r.forward(1, M, vb[37:40], W, H, s["loc"], s["sh"], s["opac"], s["scale"], 1.0, s["rot"], vb[0:16], vb[16:32],
                       vb[32:35], float(vb[35]), float(vb[36]))
    tiles = r.get("tiles_touched")
    assert int(tiles.sum()) == R == r.get("point_offsets")[-1]
    keys = r.get("keys")
    assert np.all(keys[1:] >= keys[:-1])
    # stable: equal keys keep ascending splat index
    pl = r.get("point_list")
    eq = keys[1:] == keys[:-1]
    assert np.all(pl[1:][eq] > pl[:-1][eq])
    ranges = r.get("ranges").reshape(-1, 2)
    ne = ranges[ranges[:, 1] > ranges[:, 0]]
    assert ne[0, 0] == 0 and ne[-1, 1] == R and np.all(ne[1:, 0] == ne[:-1, 1])  # partition of [0, R)
    fT = r.get("final_T")
    assert fT.min() >= 0 and fT.max() <= 1
    ncon = r.get("n_contrib").reshape(H, W)
    lens = (ranges[:, 1] - ranges[:, 0]).reshape(H // 16, W // 16)
    assert np.all(ncon <= np.repeat(np.repeat(lens, 16, 0), 16, 1))
    # culled splats get exactly-zero gradients
    g = r.backward(np.ones((3, H, W), np.float32))
    culled = r.get("radii") <= 0
    assert culled.any()
    for name, n in [("dL_dmean3D", 3), ("dL_dscale", 3), ("dL_drot", 4), ("dL_dsh", 3 * M), ("dL_dopacity", 1)]:
        assert not g[name].reshape(P, n)[culled].any()


def test_reference_image_kernels(orc):
    w, h = 5, 3
    src = np.zeros((3, h, w), np.float32)
    src[0].flat[:6] = [-0.1, 0.0, 0.5, 0.999, 1.0, 7.0]
    src[1] += 0.25
    src[2] += 255.0 / 256.0
    fb = orc.image_float_to_int(src, w, h)
    assert list(fb[:6] & 0xFF) == [0, 0, 128, 255, 255, 255]          # x256, clamped (src/Trainer.cu:25)
    assert np.all((fb >> 8) & 0xFF == 64) and np.all((fb >> 16) & 0xFF == 255) and np.all(fb >> 24 == 0xFF)
    truth = np.full(w * h, 0x11FF8000, np.uint32)                    # alpha byte ignored
    loss = orc.image_int_to_loss(truth, src, w, h).reshape(3, h, w)
    assert np.array_equal(loss[0], np.float32(0.0) / np.float32(255) - src[0])
    assert np.array_equal(loss[1], np.float32(128) / np.float32(255) - src[1])
    assert np.array_equal(loss[2], np.float32(255) / np.float32(255) - src[2])


def test_accumulate_and_apply_semantics(orc):
    import ctypes as C
    P, M = 3, 4
    rng = np.random.default_rng(0)
    g = dict(loc=rng.normal(size=3 * P), sh=rng.normal(size=3 * M * P), scale=rng.normal(size=3 * P), opac=rng.normal(size=P),
             rot=rng.normal(size=4 * P))
    g = {k: v.astype(np.float32) for k, v in g.items()}
    # applyGradients: scale clamps to [0, maxScale], opacity to [0, 1], rotation NOT renormalised
    p = dict(loc=np.zeros(3 * P, np.float32), sh=np.zeros(3 * M * P, np.float32), scale=np.full(3 * P, 0.29, np.float32),
             opac=np.full(P, 0.99, np.float32), rot=np.tile(np.array([1, 0, 0, 0], np.float32), P))
    g["scale"][:] = [5, -5, 0.1] * P
    g["opac"][:] = [5, -5, 0.001]
    orc.apply_sgd(p["loc"], p["sh"], p["scale"], p["opac"], p["rot"], g, (0.1, 0.2, 0.1, 0.1, 0.3), 0.3, M)
    assert np.array_equal(p["loc"], g["loc"] * np.float32(0.1))
    assert np.allclose(p["scale"].reshape(P, 3), [0.3, 0.0, 0.3])
    assert np.allclose(p["opac"], [1.0, 0.49, 0.9901], atol=1e-6)
    assert np.array_equal(p["rot"], np.tile(np.array([1, 0, 0, 0], np.float32), P) + g["rot"] * np.float32(0.3))
    # accumulateGradients: var += |g_loc| / S, avg += g / S  (division, src/Trainer.cu:52-75)
    L = orc.lib()
    var = np.zeros(P, np.float32)
    a = {k: np.zeros_like(v) for k, v in g.items()}
    f = lambda x: x.ctypes.data_as(C.POINTER(C.c_float))
    L.orc_accumulate(f(var), f(a["loc"]), f(a["sh"]), f(a["scale"]), f(a["opac"]), f(a["rot"]), f(g["loc"]), f(g["sh"]),
                     f(g["scale"]), f(g["opac"]), f(g["rot"]), C.c_float(6.0), C.c_int(M), C.c_int(P))
    assert np.array_equal(a["sh"], g["sh"] / np.float32(6.0))
    gl = g["loc"].reshape(P, 3)
    assert np.allclose(var, np.sqrt((gl * gl).sum(1)) / 6.0, rtol=1e-6)


def test_camera_conventions(orc):
    cams = gs.camera.get_cameras(16)
    pts = np.array([c.location for c in cams])
    assert np.allclose(np.linalg.norm(pts, axis=1), 10.0, atol=1e-4)       # Fibonacci sphere radius = distance
    assert np.allclose(pts, orc.fibonacci_sphere(16, 10.0), atol=1e-5)
    for c in cams[:4]:
        view = c.getView()
        assert np.allclose(view, orc.camera_view(c.location), atol=1e-6)
        V = view.reshape(4, 4).T
        assert np.allclose(V[3], [0, 0, 0, -1])                            # every entry of lookAt negated
        assert np.allclose((V @ np.array([0, 0, 0, 1.0]))[2], 10.0, atol=1e-4)  # +z is forward: target at depth 10
        assert np.allclose(c.getProjection(1.0), orc.camera_proj(60.0, 1.0), atol=1e-6)
        assert np.allclose(gs.camera.mat4_mul(c.getProjection(1.5), view), orc.mat4_mul(orc.camera_proj(60.0, 1.5), view), atol=1e-5)
    v = gs.camera.train_views(cams, 64, 64)
    assert v.shape == (32, 40) and np.all(v[:16, 37:] == 1) and np.all(v[16:, 37:] == 0)   # white first, then black
    assert np.allclose(v[0, 35:37], math.tan(math.radians(30)))


def _densify(orc, loc, scale, opac, rot, var, grad, hp, quat_xyzw=1, M=1, cap=8):
    n = len(opac)
    L = np.zeros(cap * 3, np.float32); L[:3 * n] = np.ravel(loc)
    S = np.zeros(cap * 3, np.float32); S[:3 * n] = np.ravel(scale)
    O = np.zeros(cap, np.float32); O[:n] = opac
    Rr = np.zeros(cap * 4, np.float32); Rr[:4 * n] = np.ravel(rot)
    SH = np.zeros(cap * 3 * M, np.float32); SH[:3 * M * n] = np.arange(3 * M * n)
    n2 = orc.densify(L, SH, S, O, Rr, n, cap, M, np.asarray(var, np.float32), np.ravel(np.asarray(grad, np.float32)), hp, quat_xyzw)
    return n2, L.reshape(cap, 3), S.reshape(cap, 3), O, Rr.reshape(cap, 4), SH.reshape(cap, 3 * M)


HP = dict(cull_opacity=0.005, cull_size=0.004, densify_variance=2.0, split_size=0.04, split_distance=1.5, split_scale=0.8,
          clone_distance=1.6)


def test_densify_split_clone_prune(orc):
    s2 = math.sqrt(0.5)
    loc = [[0, 0, 0], [1, 1, 1], [2, 2, 2], [3, 3, 3]]
    scale = [[0.1, 0.02, 0.03], [0.01, 0.01, 0.01], [0.1, 0.1, 0.1], [0.001, 0.001, 0.001]]
    opac = [0.5, 0.5, 0.001, 0.5]
    rot = [[s2, 0, 0, s2], [1, 0, 0, 0], [1, 0, 0, 0], [1, 0, 0, 0]]   # splat 0: 90 deg about z (w,x,y,z)
    var = [5, 8, 5, 5]
    grad = [[0, 0, 1], [0, 3, 4], [0, 0, 0], [0, 0, 0]]
    n2, L, S, O, Rr, SH = _densify(orc, loc, scale, opac, rot, var, grad, HP)
    # splats 2 (transparent) and 3 (tiny) pruned; 0 split (size > 0.04); 1 cloned  -> 2 + 2 = 4
    assert n2 == 4
    # split: +-0.5 * 1.5 * (R * (0.1,0,0)) = +-0.075 along +y; scales * 0.8; quaternion stored permuted {x,y,z,w}
    assert np.allclose(L[0], [0, 0.075, 0], atol=1e-6) and np.allclose(L[2], [0, -0.075, 0], atol=1e-6)
    assert np.allclose(S[0], np.array([0.1, 0.02, 0.03]) * 0.8) and np.allclose(S[2], S[0])
    assert np.allclose(Rr[0], [0, 0, s2, s2]) and np.allclose(Rr[2], Rr[0])
    # clone: original untouched, copy offset by scale (.) normalize(grad) * 1.6
    assert np.allclose(L[1], [1, 1, 1]) and np.allclose(L[3], [1, 1 + 0.01 * 0.6 * 1.6, 1 + 0.01 * 0.8 * 1.6], atol=1e-6)
    assert np.array_equal(SH[3], SH[1]) and np.array_equal(SH[2], SH[0])
    # GLM_FORCE_QUAT_DATA_WXYZ variant leaves the quaternion alone
    n2, L, S, O, Rr, SH = _densify(orc, loc, scale, opac, rot, var, grad, HP, quat_xyzw=0)
    assert np.allclose(Rr[0], [s2, 0, 0, s2])


def test_densify_respects_capacity_and_variance_gate(orc):
    loc = [[0, 0, 0], [1, 1, 1]]
    scale = [[0.1, 0.1, 0.1]] * 2
    n2, *_ = _densify(orc, loc, scale, [0.5, 0.5], [[1, 0, 0, 0]] * 2, [5, 5], [[0, 0, 1]] * 2, HP, cap=3)
    assert n2 == 3                                        # only one split fits (count < capacity guard)
    n2, *_ = _densify(orc, loc, scale, [0.5, 0.5], [[1, 0, 0, 0]] * 2, [2.5, 2.5], [[0, 0, 1]] * 2, HP)
    assert n2 == 2                                        # var - |grad| = 1.5 <= 2.0: nothing happens


def _quat_mul(a, b):   # (w, x, y, z)
    w1, x1, y1, z1 = a
    w2, x2, y2, z2 = b
    return np.array([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2])


def test_rigid_motion_of_scene_and_camera_leaves_the_image_unchanged(orc):
    """Source-independent pin of the whole projection chain (view transform, EWA Jacobian, quaternion -> covariance):
    moving splats AND camera by the same rigid transform must not change the picture.  SH degree 0 (higher degrees
    are defined in the world frame and do not rotate with the scene).  fp64 oracle: agreement to 1e-9."""
    P, M, W, H = 300, 1, 96, 64
    s = gs.synth.random_splats(P, M, 5150)
    cam = gs.camera.get_cameras(3)[1]
    vb = gs.camera.view_block(cam, W, H, white=True).astype(np.float64)
    q = np.array([0.3, -0.5, 0.7, 0.4]); q /= np.linalg.norm(q)
    w, x, y, z = q
    Rm = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                   [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                   [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    t = np.array([0.7, -1.3, 2.1])
    Tinv = np.eye(4); Tinv[:3, :3] = Rm.T; Tinv[:3, 3] = -Rm.T @ t
    col = lambda m: m.T.reshape(-1)                      # 4x4 -> glm column-major 16 floats
    mat = lambda v: np.asarray(v).reshape(4, 4).T
    view2, proj2 = col(mat(vb[0:16]) @ Tinv), col(mat(vb[16:32]) @ Tinv)
    campos2 = Rm @ vb[32:35] + t
    loc = s["loc"].reshape(P, 3).astype(np.float64)
    rot = s["rot"].reshape(P, 4).astype(np.float64)
    rot /= np.linalg.norm(rot, axis=1, keepdims=True)   # the rasterizer does not normalise: R(q r) = R(q) R(r) needs |r| = 1 exactly
    loc2 = loc @ Rm.T + t
    rot2 = np.stack([_quat_mul(q, r) for r in rot])
    imgs = []
    for L, Rq, V, Pm, C in ((loc, rot, vb[0:16], vb[16:32], vb[32:35]), (loc2, rot2, view2, proj2, campos2)):
        r = orc.Rasterizer(np.float64)
        out, _ = r.forward(0, M, vb[37:40], W, H, L, s["sh"], s["opac"], s["scale"], 1.0, Rq, V, Pm, C, float(vb[35]), float(vb[36]))
        imgs.append(out)
    assert imgs[0].std() > 0.01                          # a non-trivial picture
    assert np.abs(imgs[0] - imgs[1]).max() < 1e-9


def test_uniform_scaling_of_scene_and_camera_leaves_the_image_unchanged(orc):
    """Second source-independent pin: scaling positions, splat sizes and the camera distance by one factor k keeps
    every projected mean and every 2D covariance (J ~ 1/k, cov3D ~ k^2), hence the picture; depths scale by k, the
    sort order does not change.  Any degree of SH (directions are unchanged)."""
    P, M, W, H, k = 300, 9, 96, 64, 2.5
    s = gs.synth.random_splats(P, M, 777)
    cam = gs.camera.get_cameras(4)[2]
    vb = gs.camera.view_block(cam, W, H, white=False).astype(np.float64)
    mat = lambda v: np.asarray(v).reshape(4, 4).T
    col = lambda m: m.T.reshape(-1)
    V = mat(vb[0:16]); PV = mat(vb[16:32])
    Pm = PV @ np.linalg.inv(V)
    V2 = V.copy(); V2[:3, 3] *= k
    imgs, depths = [], []
    for L, S, Vm, C in ((s["loc"].astype(np.float64), s["scale"].astype(np.float64), V, vb[32:35]),
                        (s["loc"].astype(np.float64) * k, s["scale"].astype(np.float64) * k, V2, vb[32:35] * k)):
        r = orc.Rasterizer(np.float64)
        out, _ = r.forward(2, M, vb[37:40], W, H, L, s["sh"], s["opac"], S, 1.0, s["rot"], col(Vm), col(Pm @ Vm), C, float(vb[35]), float(vb[36]))
        imgs.append(out)
    assert imgs[0].std() > 0.01
    # not 1e-9: upstream divides by (w + 1e-7), and that epsilon does not scale with the scene (1e-8 relative at w = 10)
    assert np.abs(imgs[0] - imgs[1]).max() < 2e-6


def test_sh_basis_is_an_orthonormal_real_sh_basis(orc):
    """Source-independent pin of the colour model: with SH coefficient k set to 1/2 (others 0) the oracle's colour is
    0.5 + Y_k(direction) / 2 (never negative, so never clamped), so probing it over the whole sphere gives the sixteen
    basis functions the path uses.  They must be orthonormal under the sphere's surface measure (the real spherical
    harmonics up to sign and order: wrong constants or polynomials break this), and the degree-0 / degree-1 ones have
    the closed forms 1/(2 sqrt(pi)) and sqrt(3/(4 pi)) * {y, z, x} up to sign."""
    n = 4000
    pts = gs.camera.fibonacci_sphere(n, 3.0).astype(np.float64)          # quasi-uniform quadrature nodes, radius 3
    dirs = pts / np.linalg.norm(pts, axis=1, keepdims=True)
    M, W, H = 16, 64, 64
    Y = np.full((n, M), np.nan)
    # six cube-face cameras at the origin (tan 1.3 > 1: every direction is inside some frustum).  view = an orthonormal
    # frame whose +z is the face normal; the projection only has to put the splat inside the image.
    I3 = np.eye(3)
    faces = [np.stack([I3[(a + 1) % 3], I3[(a + 2) % 3], sgn * I3[a]]) for a in range(3) for sgn in (1.0, -1.0)]
    tan = 1.3
    col = lambda m: np.ascontiguousarray(m.T.reshape(-1), np.float32)     # glm column-major
    for Rw in faces:
        view = np.eye(4); view[:3, :3] = Rw
        proj = np.zeros((4, 4)); proj[0, 0] = 1 / tan; proj[1, 1] = 1 / tan; proj[2, 2] = 1.0; proj[3, 2] = 1.0
        pv = proj @ view
        for k in range(M):
            sh = np.zeros((n, M, 3), np.float32); sh[:, k, :] = 0.5
            r = orc.Rasterizer(np.float32)
            r.forward(3, M, np.zeros(3, np.float32), W, H, pts.astype(np.float32), sh, np.full(n, 0.5, np.float32),
                      np.full(3 * n, 0.05, np.float32), 1.0, np.tile(np.array([1, 0, 0, 0], np.float32), n), col(view), col(pv),
                      np.zeros(3, np.float32), tan, tan)
            vis = r.get("radii") > 0
            assert not r.get("clamped")[np.repeat(vis, 3)].any()
            Y[vis, k] = (r.get("rgb").reshape(n, 3)[vis, 0].astype(np.float64) - 0.5) * 2.0
    assert not np.isnan(Y).any()                                          # the six frusta cover the sphere
    gram = (4.0 * math.pi / n) * (Y.T @ Y)
    assert np.abs(gram - np.eye(M)).max() < 5e-3, np.abs(gram - np.eye(M)).max()
    assert np.allclose(Y[:, 0], 0.5 / math.sqrt(math.pi), atol=1e-6)
    c1 = math.sqrt(3.0 / (4.0 * math.pi))
    for k, axis in ((1, 1), (2, 2), (3, 0)):                               # |Y_1..3| = c1 * |y|, |z|, |x|
        assert np.allclose(np.abs(Y[:, k]), c1 * np.abs(dirs[:, axis]), atol=2e-6), k
