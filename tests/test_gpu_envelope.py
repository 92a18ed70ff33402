"""The HIP path against the reference's own run-to-run envelope (round-4 verdict, item 1).

north_star asks for gradients "within 1e-4 relative" of the reference.  The reference's backward is not a function: its nine
pixel-stage sums are fp32 atomicAdds in hardware order (src/Trainer.cu:378-412), so "the reference" is a set of admissible runs.
oracle/gs_oracle.cpp (atomic_prepare / atomic_sums) emulates that arithmetic — the same fp32 terms, added in fp32 in a seeded
order, then the unchanged chain and accumulateGradients — and tests/test_reference_noise.py pins the emulation on the CPU.  Here
K = 8 such runs give every gradient entry an envelope [min, max], and the HIP step (per-pass form and fused-pair form) and the
rasterizer seam are held to THAT: util.envelope_verdict classifies every entry as
    0  within 1e-4 of the value itself,   1  inside the 16 x widened envelope,   2  within 256 x 2^-24 of sum|term| (entries whose K
    runs agree bit for bit: no order noise to compare with),   3  a NAMED decision flip inside the old accounted budget,   4  unexplained
— with NO conditioning, chain-noise or A-noise term — and the tests assert zero unexplained entries, classes 2 and 3 rare, and, where
splats have enough terms for the statistics to mean something (cfg2, cfg3), that the HIP result is inside the 4 x widened envelope
as often as a held-out run of the reference itself (minus 3 points).  Measured on the MI355X (profiles/r05/parity_table.md): at
cfg3 the HIP gradients are inside the plain envelope for 75-82 % of the entries, a held-out reference run for 80 % (7/9 expected)."""
import numpy as np
import pytest

import gsplat_amd as gs
import util
from util import envelope_verdict, make_scene, step_budget, view_parts

pytestmark = pytest.mark.gpu
K = 8


def _assert_rates(label, k, r, statistical, flip_share=2e-4, ulps_share=2e-4):
    n = r["n"]
    assert r["unexplained"] == 0, f"{label} / {k}: {r['unexplained']} unexplained of {n}; the worst: {r.get('worst_unexplained')}"
    assert r["by_flip"] <= max(4, flip_share * n), (label, k, r)       # measured at cfg3: 89 of 6.0 M (1.5e-5)
    assert r["by_ulps"] <= max(4, ulps_share * n), (label, k, r)       # measured: 2 of 33 600, 1 of 4.8 M; 84 of 180 000 where pixels blend 3204 entries
    if statistical:
        assert r["within4"] >= r["ref_within4"] - 0.03, (label, k, r)


def _line(label, form, rates):
    cols = []
    for k, r in rates.items():
        cols.append(f"{k}: strict {r['strict']:.4f}, inside {r['inside']:.3f} (ref {r['ref_inside']:.3f}), 4x {r['within4']:.4f} (ref {r['ref_within4']:.4f}), "
                    f"16x {r['within16']:.5f}, ulps {r['by_ulps']}, flips {r['by_flip']}, unexplained {r['unexplained']} of {r['n']}")
    print(f"[envelope, {label}, {form}] " + "; ".join(cols))


@pytest.mark.parametrize("P,M,n_cams,W,H", [(1000, 4, 1, 256, 256),      # BASELINE cfg1
                                              (1500, 1, 3, 128, 96),
                                              (700, 16, 2, 112, 112),
                                              (10000, 1, 4, 512, 512),     # BASELINE cfg2
                                              (100000, 16, 8, 1024, 1024)])  # BASELINE cfg3
def test_step_gradients_lie_in_the_references_own_envelope(orc, P, M, n_cams, W, H):
    import test_gpu_trainer as tg
    s, cams, fw, fb, tr = tg._setup(orc, P, M, n_cams, W, H, 0x5EED0001)
    views = gs.camera.train_views(cams, W, H)
    truths = np.concatenate(fw + fb)
    st = tr.accumulate(stats=True)
    g = tg._read_grads(tr, P, M)
    proj = gs.Project()
    tr.apply(proj)
    host = gs.ModelSplatsHost.fromVectors(s["loc"], s["sh"], s["scale"], s["opac"], s["rot"])
    host.shDegree = s["D"]
    tr.model = gs.ModelSplatsDevice(host)
    tr.train(proj, densify=False)
    gf = tg._read_grads(tr, P, M)
    tr.close()
    bud = step_budget(orc, s, s["D"], M, W, H, views, truths, 2.0 * n_cams, flip_margin=1e-4 if st.max_tile_list <= 1024 else 1e-3, atomic_seeds=range(K))
    stride = dict(loc=3, sh=3 * M, scale=3, opac=1, rot=4, var=1)
    label = f"{P} splats, {2 * n_cams} passes @{W}x{H}"
    statistical = P >= 10000
    for form, got in (("per-pass form", g), ("fused-pair step", gf)):
        rates = {}
        for k in ["loc", "sh", "scale", "opac", "rot", "var"]:
            if form != "per-pass form" and k == "var":
                continue       # `var` has no reader on a step without densify and is zero there (tests/test_gpu_trainer.py)
            _, rates[k] = envelope_verdict(got[k], bud[k]["want"], bud["runs"][k], bud[k]["sumabs"], bud[k]["flip"], bud[k]["budget"], stride[k])
        _line(label, form, rates)
        for k, r in rates.items():
            _assert_rates(label, k, r, statistical)


SEAM_NAMES = [("dL_dmean3D", 3), ("dL_dcov3D", 6), ("dL_dsh", None), ("dL_dscale", 3), ("dL_drot", 4)]


@pytest.mark.parametrize("P,M,D,W,H,seed", [(1000, 4, 1, 256, 256, 0x5EED0001), (800, 1, 0, 250, 130, 11), (600, 9, 2, 96, 160, 12),
                                             (500, 16, 3, 128, 128, 13), (20000, 1, 0, 200, 72, 14)])       # tests/test_gpu_raster.py::CASES
def test_seam_backward_lies_in_the_references_own_envelope(orc, P, M, D, W, H, seed):
    s, cams, views = make_scene(P, M, seed, W, H, n_cams=2)
    vp = view_parts(views[1])
    sr = util.SeamRaster()
    sr.forward(s, D, M, vp, W, H)
    r, _, _ = util.oracle_forward(orc, s, D, M, vp, W, H)
    dpix = np.random.default_rng(seed).uniform(-1, 1, (3, H, W)).astype(np.float32)
    g = sr.backward(dpix)
    og = r.backward(dpix, want_abs=True, flip_margin=1e-4)
    r.atomic_prepare(dpix)
    runs = [r.atomic_backward(sd, 0) for sd in range(K)]
    r.atomic_release()
    nine = lambda d: np.concatenate([d["dL_dcolor"].reshape(P, 3), d["dL_dmean2D"].reshape(P, 3)[:, :2], d["dL_dconic"].reshape(P, 4)[:, [0, 1, 3]],
                                     d["dL_dopacity"].reshape(P, 1)], axis=1)
    rates = {}
    abs9, flip9 = og["abs9"], og["flip9"]
    # T is a running product of one factor per blended entry: a pixel that blends n entries hands every term n ulps of relative error whatever
    # the order of the factors (the forward multiplies front to back, the backward divides back to front) — the 20 000-splat scene blends up
    # to 3204 entries per pixel.  The ulps class grows with that count beyond 1024 entries; the four other scenes stay at 256.
    ulps = util.ENVELOPE_ULPS * max(1.0, float(r.get("n_contrib").max()) / 1024.0)
    _, rates["nine sums"] = envelope_verdict(nine(g), nine(og), [nine(x) for x in runs], abs9, flip9, 1e-4 * (abs9 + flip9) + flip9, 9, ulps=ulps)
    # chain outputs: sum|term| and the flip part carried through the chain (|A| = the chain on the nine unit inputs, as tests/test_gpu_raster.py)
    names = [(n, k if k else 3 * M) for n, k in SEAM_NAMES]
    sa = {n: np.zeros((P, k)) for n, k in names}
    fl = {n: np.zeros((P, k)) for n, k in names}
    for q in range(9):
        unit = np.zeros((P, 9), np.float32); unit[:, q] = 1.0
        col = orc.chain(r, unit)
        for n, k in names:
            A = np.abs(col[n].reshape(P, k).astype(np.float64))
            sa[n] += A * abs9[:, q, None]; fl[n] += A * flip9[:, q, None]
    for n, k in names:
        _, rates[n] = envelope_verdict(g[n], og[n], [x[n] for x in runs], sa[n], fl[n], 1e-4 * (sa[n] + fl[n]) + fl[n], k, ulps=ulps)
    label = f"seam {P} splats @{W}x{H}, M={M}" + (f", ulps class x{ulps / util.ENVELOPE_ULPS:.1f}" if ulps > util.ENVELOPE_ULPS else "")
    _line(label, "gs_rasterize_backward", rates)
    # (one flipped pixel under a uniform random dL/dpixel moves the sums of every splat blended there: 26 of the 7200 sums of the
    #  800-splat scene, the one scene of the five with a flipped pixel — tests/test_gpu_raster.py allows the flip allowance on 5 % of the splats)
    for k, rr in rates.items():
        _assert_rates(label, k, rr, False, flip_share=0.01, ulps_share=2e-3 if ulps > util.ENVELOPE_ULPS else 2e-4)   # (measured in the deep scene: up to 6.5e-4 of an array)
