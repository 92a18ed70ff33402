"""BASELINE.json's FULL sizes (cfg2: 10k splats, 8 passes @512^2; cfg3: 100k splats, 16 passes @1024^2, SH degree 3;
for the tile lists also cfg5: 1M splats @2048^2) on the GPU, checked through properties that do not need the oracle
to run at that size:

* the sorted tile lists ARE a stable sort of (tile, depth) over the emitted (splat, tile) pairs (sortedness, stability,
  multiset equality with tiles_touched, ranges partition [0, R));
* white-background image minus black-background image = final_T, exactly the blend identity of Appendix A.6;
* the backward is linear in dL/dpixel: doubling the input doubles every output BIT FOR BIT (power-of-two scaling is
  exact in fp32), through the same reductions;
* culling on/off, camera-pass sharing on/off and repeated runs agree bit for bit; the split API equals the step bit for
  bit in the per-pass form and up to re-association of the pass sums in the fused-pair form;
* the model survives upload -> download unchanged, a clone equals its source.
"""
import ctypes as C

import numpy as np
import pytest

import gsplat_amd as gs
from gsplat_amd import capi
from util import REC_DTYPE, SeamRaster, make_scene, step_budget, unexplained, view_parts

pytestmark = pytest.mark.gpu


def _cfg(idx):
    P, M, V, W, H = gs.synth.CONFIGS[idx]
    D = gs.synth.sh_degree_for(M)
    s = gs.synth.random_splats(P, M, gs.synth.seed_for(idx))
    cams = gs.camera.get_cameras(max(V // 2, 1))
    return P, M, D, V, W, H, s, cams


def _product_truths(s_truth, D, cams, W, H):
    """Truth images rendered by the product's own preview path (content is irrelevant to these properties)."""
    host = gs.ModelSplatsHost.fromVectors(s_truth["loc"], s_truth["sh"], s_truth["scale"], s_truth["opac"], s_truth["rot"])
    host.shDegree = D
    tr = gs.Trainer(W, H)
    tr.model = gs.ModelSplatsDevice(host)
    fw = [tr.render(W, H, 1.0, c, background=(1.0, 1.0, 1.0)).reshape(-1) for c in cams]
    fb = [tr.render(W, H, 1.0, c, background=(0.0, 0.0, 0.0)).reshape(-1) for c in cams]
    tr.close()
    return fw, fb


def _trainer(idx):
    P, M, D, V, W, H, s, cams = _cfg(idx)
    t = gs.synth.random_splats(max(P // 2, 1), M, gs.synth.seed_for(idx) + 1000)
    fw, fb = _product_truths(t, D, cams, W, H)
    host = gs.ModelSplatsHost.fromVectors(s["loc"], s["sh"], s["scale"], s["opac"], s["rot"])
    host.shDegree = D
    tr = gs.Trainer(W, H)
    tr.model = gs.ModelSplatsDevice(host)
    tr.captureTruths(cams, fw, fb)
    return tr, s, P, M


def _download(tr):
    h = gs.ModelSplatsHost.fromDevice(tr.model)
    return dict(loc=h.locations.copy(), sh=h.shs.copy(), scale=h.scales.copy(), opac=h.opacities.copy(), rot=h.rotations.copy())


def _grads(tr):
    ptr, n = tr.grad_buffer()
    tr.synchronize()
    buf = np.empty(n, np.float32)
    capi.check(capi.lib().gs_memcpy_d2h(buf.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), n * 4))
    return buf


@pytest.mark.parametrize("idx", [2, 3, 5])   # 5: 1M splats @2048^2 — ~1100 entries per tile, lists beyond the LDS sort (spill path)
def test_lists_are_the_stable_sort_and_blend_identity(idx):
    P, M, D, V, W, H, s, cams = _cfg(idx)
    views = gs.camera.train_views(cams, W, H)
    n_cams = len(cams)
    vw, vb = view_parts(views[1]), view_parts(views[n_cams + 1])   # camera 1, white and black background
    _check_lists_and_blend_identity(s, P, M, D, W, H, vw, vb)


def _check_lists_and_blend_identity(s, P, M, D, W, H, vw, vb):
    sr = SeamRaster()
    out_w, R = sr.forward(s, D, M, vw, W, H)
    tt = sr.field("geometry", "tiles_touched", np.uint32)
    po = sr.field("geometry", "point_offsets", np.uint32)
    rec = sr.field("geometry", "record", np.uint8).view(REC_DTYPE)
    pl = sr.field("binning", "point_list", np.uint32)
    ranges = sr.field("image", "ranges", np.uint32).reshape(-1, 2).astype(np.int64)
    assert R > 0 and int(tt.sum(dtype=np.int64)) == R == int(po[-1])
    assert np.array_equal(np.cumsum(tt, dtype=np.uint64).astype(np.uint32), po)
    # ranges: non-empty tiles partition [0, R) in tile order
    ne = ranges[ranges[:, 1] > ranges[:, 0]]
    assert ne[0, 0] == 0 and ne[-1, 1] == R and np.array_equal(ne[1:, 0], ne[:-1, 1])
    # every splat appears once per tile it touches, inside its rectangle
    assert np.array_equal(np.bincount(pl, minlength=P).astype(np.uint32), tt)
    gx = (W + 15) // 16
    lens = (ranges[:, 1] - ranges[:, 0]).clip(min=0)
    tile_of_entry = np.repeat(np.arange(len(ranges)), lens)
    tx, ty = tile_of_entry % gx, tile_of_entry // gx
    r0, r1 = rec["rect_min"][pl], rec["rect_max"][pl]
    assert np.all((tx >= (r0 & 0xffff)) & (tx < (r1 & 0xffff)) & (ty >= (r0 >> 16)) & (ty < (r1 >> 16)))
    # sorted by (tile, depth bits), ties in ascending splat id: upstream's stable radix sort of (tile << 32 | depth)
    key = (tile_of_entry.astype(np.uint64) << np.uint64(32)) | rec["depth"][pl].view(np.uint32).astype(np.uint64)
    assert np.all(key[1:] >= key[:-1])
    tie = key[1:] == key[:-1]
    assert np.all(pl[1:][tie] > pl[:-1][tie])
    fT = sr.field("image", "final_T", np.float32)
    ncon = sr.field("image", "n_contrib", np.uint32)
    assert fT.min() >= 0.0 and fT.max() <= 1.0
    px_tile = (np.arange(H)[:, None] // 16) * gx + (np.arange(W)[None, :] // 16)
    assert np.all(ncon.reshape(H, W) <= lens[px_tile])
    # blend identity: the two backgrounds differ by exactly T per channel (fp32: one rounding of C + T*bg)
    sb = SeamRaster()
    out_b, Rb = sb.forward(s, D, M, vb, W, H)
    assert Rb == R and np.array_equal(sb.field("binning", "point_list", np.uint32), pl)
    for c in range(3):
        assert np.abs((out_w[c] - out_b[c]).reshape(-1) - fT).max() <= 2e-7 * max(1.0, float(out_w[c].max()))


@pytest.mark.parametrize("W,H,P", [(8192, 8192, 20000), (8192, 16, 20000)])
def test_render_at_the_reference_maximum_size(orc, W, H, P):
    """The reference's "Render Splats" tool offers up to 8192 x 8192 (src/ui/tools/UiPanelToolsView.cpp:120,125; call at
    :249-250): 16384 super-tiles of 64 x 64 px, 262144 tiles (the three-phase tile scan, one LDS counter per super-tile =
    64 KB in the projection and the coarse scatter).  The lists are the stable sort at that size too, and Trainer::render
    delivers imageFloatToInt of the very picture the rasterizer seam computes."""
    M, D = 4, 1
    s = gs.synth.random_splats(P, M, 8192 + H)
    cam = gs.camera.get_cameras(3)[1]
    import math
    blk_w = gs.camera.view_block(cam, W, H, white=True)
    blk_b = gs.camera.view_block(cam, W, H, white=False)
    if H == 16:   # a 512 : 1 strip: keep the horizontal field sane (the reference's tan_fovx formula is its caller's business)
        for b in (blk_w, blk_b):
            b[35] = np.float32(1.0)
    _check_lists_and_blend_identity(s, P, M, D, W, H, view_parts(blk_w), view_parts(blk_b))
    # Trainer::render (src/Trainer.cu:148-216) on the same pass parameters
    host = gs.ModelSplatsHost.fromVectors(s["loc"], s["sh"], s["scale"], s["opac"], s["rot"])
    host.shDegree = D
    tr = gs.Trainer(W, H)
    tr.model = gs.ModelSplatsDevice(host)
    v = capi.view_from_block(blk_b)
    fb = np.zeros(W * H, np.uint32)
    capi.check(capi.lib().gs_trainer_render(tr.handle, fb.ctypes.data_as(C.c_void_p), 0, W, H, C.c_float(1.0), C.byref(v)))
    tr.close()
    out_b, _ = SeamRaster().forward(s, D, M, view_parts(blk_b), W, H)
    assert np.array_equal(fb, orc.image_float_to_int(out_b, W, H))
    assert len(np.unique(fb)) > 100   # a picture, not a constant


def test_backward_is_exactly_linear_under_doubling_cfg3():
    P, M, D, V, W, H, s, cams = _cfg(3)
    views = gs.camera.train_views(cams, W, H)
    sr = SeamRaster()
    sr.forward(s, D, M, view_parts(views[0]), W, H)
    rng = np.random.default_rng(5)
    dpix = rng.uniform(-1, 1, (3, H, W)).astype(np.float32)
    g1 = sr.backward(dpix)
    g2 = sr.backward(2.0 * dpix)
    again = sr.backward(dpix)
    for k in g1:
        assert np.array_equal(g1[k].view(np.uint32), again[k].view(np.uint32)), k            # reproducible
        assert np.array_equal((2.0 * g1[k]).view(np.uint32), g2[k].view(np.uint32)), k       # linear, exactly
    assert np.abs(g1["dL_dmean3D"]).max() > 0 and np.isfinite(g1["dL_dsh"]).all()


@pytest.mark.parametrize("option", ["cull", "share_camera_passes"])
def test_switches_do_not_change_a_bit_at_cfg3(option):
    res = []
    # without camera-pass sharing there are no pairs to fuse: compare the per-pass form on both sides of that switch
    fuse = 0 if option == "share_camera_passes" else 1
    capi.check(capi.lib().gs_set_option(b"fuse_camera_passes", fuse))
    for value in (1, 0):
        capi.check(capi.lib().gs_set_option(option.encode(), value))
        try:
            tr, s, P, M = _trainer(3)
            st = tr.train(gs.Project(updateRule=capi.GS_UPDATE_ADAM), stats=True)
            res.append((st.num_rendered, st.max_tile_list, st.loss, _grads(tr), _download(tr)))
            tr.close()
        finally:
            capi.check(capi.lib().gs_set_option(option.encode(), 1))
    capi.check(capi.lib().gs_set_option(b"fuse_camera_passes", 1))
    assert res[0][:3] == res[1][:3]
    assert np.array_equal(res[0][3].view(np.uint32), res[1][3].view(np.uint32))
    for k in res[0][4]:
        assert np.array_equal(res[0][4][k].view(np.uint32), res[1][4][k].view(np.uint32)), k


def _three_steps(mode, proj):
    tr, s, P, M = _trainer(3)
    for _ in range(3):
        if mode == "step":
            tr.train(proj)
        else:
            tr.accumulate()
            tr.apply(proj)
    m = _download(tr)
    tr.close()
    return m, s


def test_step_is_reproducible_and_equals_the_split_api_cfg3():
    adam = gs.Project(updateRule=capi.GS_UPDATE_ADAM)
    # the step (one backward per camera pair) is bitwise reproducible
    a, s = _three_steps("step", adam)
    b, _ = _three_steps("step", adam)
    for k in a:
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k
    assert not np.array_equal(a["loc"], s["loc"].reshape(-1))   # the steps did move the model
    # with pair fusion off the step IS accumulate + apply, bit for bit
    capi.check(capi.lib().gs_set_option(b"fuse_camera_passes", 0))
    try:
        c, _ = _three_steps("step", adam)
    finally:
        capi.check(capi.lib().gs_set_option(b"fuse_camera_passes", 1))
    d, _ = _three_steps("split", adam)
    for k in c:
        assert np.array_equal(c[k].view(np.uint32), d[k].view(np.uint32)), k
    # and the fused step differs from the per-pass form only by re-association of the two-pass sums (reference rule:
    # the update is linear in the gradient, so the comparison is meaningful element by element)
    sgd = gs.Project()
    e, _ = _three_steps("step", sgd)
    f, _ = _three_steps("split", sgd)
    for k in e:
        start = np.asarray(s[k], np.float32).reshape(-1)
        moved = np.abs(f[k][:start.size] - start).max()
        assert np.abs(e[k] - f[k]).max() <= 1e-4 * moved + 1e-12, (k, np.abs(e[k] - f[k]).max(), moved)


@pytest.mark.parametrize("idx", [2, 3])
def test_model_round_trip_full_size(idx):
    P, M, D, V, W, H, s, cams = _cfg(idx)
    host = gs.ModelSplatsHost.fromVectors(s["loc"], s["sh"], s["scale"], s["opac"], s["rot"])
    dev = gs.ModelSplatsDevice(host)
    clone = gs.ModelSplatsDevice(dev)
    for d in (dev, clone):
        back = gs.ModelSplatsHost.fromDevice(d)
        assert back.count == P and back.shCoeffs == M
        for name, want in [("locations", s["loc"]), ("shs", s["sh"]), ("scales", s["scale"]), ("opacities", s["opac"]), ("rotations", s["rot"])]:
            got = getattr(back, name)[:want.size]
            assert np.array_equal(np.asarray(got).view(np.uint32), np.asarray(want, np.float32).reshape(-1).view(np.uint32)), name


def test_cfg5_per_rank_load_with_fp16_sh(orc):
    """BASELINE config 5 as it is written — 1M splats @2048^2, SH degree 3 with fp16 SH storage (trainer option "sh_fp16"),
    LDS long-list / spill sort path — at the load ONE rank of its 8-GPU run carries: 4 of the 32 cameras = 8 passes
    (scale anchor /root/reference/src/Config.h:17, SPLATS_LIMIT 1000000).
      * the step against the oracle on the half-rounded coefficients, every entry of every averaged gradient accounted for
        (util.step_budget; the long lists take the wide flip margin), in the per-pass form incl. `var`;
      * the fused-pair step: same accounting; bit-reproducible; `var` zero;
      * geometry, tile lists and statistics bit-identical with the fp32 mode (SH does not enter them)."""
    import time
    P, M, D, V, W, H, s, cams32 = _cfg(5)
    n_cams = 4
    cams = cams32[:n_cams]   # rank 0's share under camera sharding (camera c -> rank c % 8 holds c = 0, 8, 16, 24: any four do)
    t = gs.synth.random_splats(P // 2, M, gs.synth.seed_for(5) + 1000)
    fw, fb = _product_truths(t, D, cams, W, H)
    host = gs.ModelSplatsHost.fromVectors(s["loc"], s["sh"], s["scale"], s["opac"], s["rot"])
    host.shDegree = D
    tr = gs.Trainer(W, H)
    tr.set_option("sh_fp16", 1)
    tr.model = gs.ModelSplatsDevice(host)
    tr.captureTruths(cams, fw, fb)
    views = gs.camera.train_views(cams, W, H)
    truths = np.concatenate(fw + fb)
    rounded = dict(s, sh=s["sh"].astype(np.float16).astype(np.float32))
    st = tr.accumulate(stats=True)
    from test_gpu_trainer import _read_grads
    g = _read_grads(tr, P, M)
    t0 = time.time()
    bud = step_budget(orc, rounded, D, M, W, H, views, truths, 2.0 * n_cams, flip_margin=1e-3)   # the oracle's gradients and their budgets
    t_budget = time.time() - t0
    o = {k: bud[k]["want"] for k in ("loc", "sh", "scale", "opac", "rot", "var")}
    assert st.views == 2 * n_cams and st.num_rendered == int(bud["num_rendered"].sum()) and st.max_tile_list > 2048
    stride = dict(loc=3, sh=3 * M, scale=3, opac=1, rot=4, var=1)
    report = []
    for k in ["loc", "sh", "scale", "opac", "rot", "var"]:
        n_bad, worst = unexplained(k, g[k], o[k], bud[k]["budget"], stride[k])
        report.append(f"{k} {n_bad} (worst {worst:.2f})")
        assert n_bad == 0, (k, n_bad, worst)
    print(f"[cfg5 per-rank load: {P} splats, fp16 SH, {2 * n_cams} passes @{W}x{H}, {st.num_rendered / (2 * n_cams):.3g} entries per pass, longest list "
          f"{st.max_tile_list}] unexplained entries per-pass form: " + ", ".join(report) + f"  (oracle + budget {t_budget:.0f} s)")
    # the step (fused pairs): same accounting, reproducible
    res = []
    for _ in range(2):
        stf = tr.train(gs.Project(lrLocation=0.0, lrSh=0.0, lrScale=0.0, lrOpacity=0.0, lrRotation=0.0), stats=True)   # gradients only
        res.append(_read_grads(tr, P, M))
    assert stf.num_rendered == st.num_rendered
    for k in ["loc", "sh", "scale", "opac", "rot"]:
        assert np.array_equal(res[0][k].view(np.uint32), res[1][k].view(np.uint32)), k
        n_bad, worst = unexplained(k + " (fused pair)", res[0][k], o[k], bud[k]["budget"], stride[k])
        assert n_bad == 0, (k, n_bad, worst)
    assert not res[0]["var"].any()
    tr.close()
    # fp32 mode: identical lists and statistics (SH never enters geometry)
    tr32 = gs.Trainer(W, H)
    tr32.model = gs.ModelSplatsDevice(host)
    tr32.captureTruths(cams, fw, fb)
    st32 = tr32.accumulate(stats=True)
    assert (st32.num_rendered, st32.max_tile_list) == (st.num_rendered, st.max_tile_list)
    tr32.close()
