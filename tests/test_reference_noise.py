"""The reference's own run-to-run noise, and the parity budget against it (CPU; round-4 verdict, item 1).

Upstream's backward render kernel adds the nine per-(pixel, splat) terms to the splat's accumulators with fp32 atomicAdd
(the call at src/Trainer.cu:378-412; buffers pre-zeroed at :366-375): the ORDER of a splat's additions is the hardware's, so two
runs of the reference differ by the rounding of two fp32 summation orders.  The oracle's backward sums in double (it has no
order); oracle/gs_oracle.cpp atomic_prepare / atomic_sums emulate the reference's arithmetic itself — the same fp32 terms added
in fp32 in a seeded order, then the unchanged per-splat chain and accumulateGradients.  K seeds = K admissible runs of the
reference; their per-entry [min, max] is the envelope a second run of the reference lands in.

Asserted here, on the five rasterizer-seam scenes, BASELINE cfg1 and cfg2:
  (a) the budget the GPU tests hold the HIP path to (util.step_budget: 1e-4 of sum|term| carried through the chain, WITHOUT its
      decision-flip part — a re-run of the reference flips no decision, its forward is deterministic) contains every run: it is
      not tighter than the reference's own noise;
  (b) how much wider than the envelope it is (printed per array; the medians are three orders of magnitude) — i.e. "within
      1e-4 of sum|term|" is a LOOSE statement next to the reference's reproducibility, which is why tests/test_gpu_envelope.py
      holds the HIP path to the envelope itself;
  (c) the envelope does not hinge on the family of orders: one random order over all of a splat's terms, random tile order with
      random pixel order inside each tile, and the two raster orders give half-widths of the same size."""
import numpy as np
import pytest

import gsplat_amd as gs
import util
from util import make_scene, oracle_forward, view_parts

K = 8
SEAM = [(1000, 4, 1, 256, 256, 0x5EED0001), (800, 1, 0, 250, 130, 11), (600, 9, 2, 96, 160, 12), (500, 16, 3, 128, 128, 13),
        (20000, 1, 0, 200, 72, 14)]        # tests/test_gpu_raster.py::CASES
STEP = [(1000, 4, 1, 256, 256), (10000, 1, 4, 512, 512)]   # BASELINE cfg1, cfg2 (tests/test_gpu_trainer.py)


def _truths(orc, P2, M, seed, cams, W, H):
    t = gs.synth.random_splats(P2, M, seed)
    views = gs.camera.train_views(cams, W, H)
    Cn = len(cams)
    fw, fb = [], []
    for v in range(2 * Cn):
        r, out, _ = oracle_forward(orc, t, t["D"], M, view_parts(views[v]), W, H)
        (fw if v < Cn else fb).append(orc.image_float_to_int(out, W, H))
    return fw, fb


@pytest.mark.parametrize("P,M,D,W,H,seed", SEAM)
def test_seam_sums_of_every_reference_run_lie_inside_the_budget(orc, P, M, D, W, H, seed):
    s, cams, views = make_scene(P, M, seed, W, H, n_cams=2)
    vp = view_parts(views[1])
    r, _, _ = oracle_forward(orc, s, D, M, vp, W, H)
    dpix = np.random.default_rng(seed).uniform(-1, 1, (3, H, W)).astype(np.float32)
    og = r.backward(dpix, want_abs=True)
    n_terms = r.atomic_prepare(dpix)
    want = np.zeros((P, 9))
    want[:, 0:3] = og["dL_dcolor"].reshape(P, 3); want[:, 3:5] = og["dL_dmean2D"].reshape(P, 3)[:, :2]
    want[:, 5:8] = og["dL_dconic"].reshape(P, 4)[:, [0, 1, 3]]; want[:, 8] = og["dL_dopacity"]
    runs = {m: np.stack([r.atomic_sums(sd, m) for sd in range(K)]).astype(np.float64) for m in (0, 1)}
    runs[2] = np.stack([r.atomic_sums(0, 2), r.atomic_sums(0, 3)]).astype(np.float64)
    r.atomic_release()
    budget = 1e-4 * og["abs9"]
    for m, a in runs.items():
        dev = np.abs(a - want).max(0)
        # (a) the double-summed value is the correctly rounded sum of the same terms: every fp32 order lands within the budget of it
        assert (dev <= budget + 1e-30).all(), (m, float((dev / (budget + 1e-30)).max()))
    hw = {m: (a.max(0) - a.min(0)) / 2 for m, a in runs.items()}
    live = (hw[0] > 0) & (hw[1] > 0)
    ratio_budget = np.median(budget[live] / hw[0][live])
    ratio_modes = np.median(hw[1][live] / hw[0][live])
    print(f"[reference noise, seam {P} splats @{W}x{H}] {n_terms} atomicAdd terms; median budget / envelope half-width {ratio_budget:.0f}; "
          f"tile-blocked orders / fully random orders {ratio_modes:.2f}; entries whose {K} runs agree bit for bit {float((hw[0] == 0).mean()):.3f}")
    assert 20.0 <= ratio_budget <= 1e6            # (b) loose, and by how much
    assert 0.5 <= ratio_modes <= 2.0              # (c)
    # the raster orders are admissible runs too: inside the envelope of the random ones widened by its own width
    lo, hi = runs[0].min(0), runs[0].max(0)
    far = util.envelope_distance(runs[2][0].reshape(-1), lo.reshape(-1), hi.reshape(-1)) > 8 * hw[0].reshape(-1) + 1e-6 * og["abs9"].reshape(-1)
    assert far.mean() <= 1e-3, far.mean()


@pytest.mark.parametrize("P,M,n_cams,W,H", STEP)
def test_step_gradients_of_every_reference_run_lie_inside_the_budget(orc, P, M, n_cams, W, H):
    s = gs.synth.random_splats(P, M, 0x5EED0001)
    cams = gs.camera.get_cameras(n_cams)
    fw, fb = _truths(orc, max(P // 2, 1), M, 0x5EED0001 + 1000, cams, W, H)
    views = gs.camera.train_views(cams, W, H)
    bud = util.step_budget(orc, s, s["D"], M, W, H, views, np.concatenate(fw + fb), 2.0 * n_cams, atomic_seeds=range(K))
    stride = dict(loc=3, sh=3 * M, scale=3, opac=1, rot=4, var=1)
    report = []
    for k in ["loc", "sh", "scale", "opac", "rot", "var"]:
        runs = bud["runs"][k].astype(np.float64)
        want = bud[k]["want"].astype(np.float64)
        plain = 1e-4 * bud[k]["sumabs"]     # the budget without its decision-flip part
        worst = 0.0
        for j in range(K):
            n_bad, w = util.unexplained(k, runs[j], want, plain, stride[k])
            assert n_bad == 0, (k, j, n_bad, w)          # (a)
            worst = max(worst, w)
        lo, hi = util.envelope(runs)
        hw = (hi - lo) / 2
        live = hw > 0
        ratio = float(np.median(plain[live] / hw[live]))
        report.append(f"{k}: worst run at {worst:.4f} of the budget, median budget / half-width {ratio:.0f}")
        assert 20.0 <= ratio <= 1e6, (k, ratio)          # (b)
    print(f"[reference noise, {P} splats, {2 * n_cams} passes @{W}x{H}] " + "; ".join(report))


def test_atomic_sums_are_the_same_terms_in_another_order(orc):
    """The emulation adds exactly the terms the double-summing backward adds: with one term per splat and sum the order cannot
    matter and the fp32 result equals the double one; and the sum of a splat's terms in double equals render_backward's."""
    P, M, D, W, H = 300, 4, 1, 64, 64
    s, cams, views = make_scene(P, M, 5, W, H)
    r, _, _ = oracle_forward(orc, s, D, M, view_parts(views[0]), W, H)
    dpix = np.random.default_rng(0).uniform(-1, 1, (3, H, W)).astype(np.float32)
    og = r.backward(dpix, want_abs=True)
    r.atomic_prepare(dpix)
    a = np.stack([r.atomic_sums(sd, 0) for sd in range(4)] + [r.atomic_sums(0, 2), r.atomic_sums(0, 3)]).astype(np.float64)
    g = r.atomic_backward(1, 0)
    r.atomic_release()
    want = np.zeros((P, 9))
    want[:, 0:3] = og["dL_dcolor"].reshape(P, 3); want[:, 3:5] = og["dL_dmean2D"].reshape(P, 3)[:, :2]
    want[:, 5:8] = og["dL_dconic"].reshape(P, 4)[:, [0, 1, 3]]; want[:, 8] = og["dL_dopacity"]
    assert np.abs(a - want).max() <= 2e-6 * og["abs9"].max()
    assert (np.abs(a.mean(0) - want) <= 3e-7 * og["abs9"] + 1e-30).all()
    idle = og["abs9"].max(1) == 0          # splats no pixel blends: no term, exact zeros
    assert idle.any() and not a[:, idle].any()
    # atomic_backward = those sums through the unchanged chain (orc.chain), in backward()'s layout
    ch = orc.chain(r, g["sums9"])
    for n in ("dL_dmean3D", "dL_dsh", "dL_dscale", "dL_drot"):
        assert np.array_equal(ch[n].view(np.uint32), g[n].view(np.uint32))
    assert np.array_equal(g["dL_dopacity"], g["sums9"][:, 8])
