#!/usr/bin/env python3
"""Generates tests/golden/cfg1.npz (and seam_cases.npz, budget_case.npz, below) from the CPU oracle (oracle/gs_oracle.cpp) at BASELINE cfg1 scale:
1k random-init splats (seed 0x5EED0001), one camera, white + black pass, 256x256, truth = quantised
oracle render of the second splat set (seed + 1000, P/2 splats).

The reference itself holds no fixtures and cannot be run here, so these vectors pin the
ORACLE's behaviour over time (tests/test_golden.py, CPU) and give the GPU path a committed target
(tests/test_gpu_golden.py); they are not reference outputs.  Run from the repo root:
    python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import gsplat_amd as gs  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402


def build():
    P, M, V, W, H = gs.synth.CONFIGS[1]
    seed = gs.synth.seed_for(1)
    s = gs.synth.random_splats(P, M, seed)
    t = gs.synth.random_splats(P // 2, M, seed + 1000)
    cams = gs.camera.get_cameras(1)
    views = gs.camera.train_views(cams, W, H)
    out = dict(P=P, M=M, W=W, H=H, seed=seed, views=views)
    truths = []
    for v in range(2):
        b = views[v]
        r = orc.Rasterizer(np.float32)
        img, _ = r.forward(1, M, b[37:40], W, H, t["loc"], t["sh"], t["opac"], t["scale"], 1.0, t["rot"], b[0:16], b[16:32], b[32:35],
                           float(b[35]), float(b[36]))
        truths.append(orc.image_float_to_int(img, W, H))
    out["truth_sha256"] = np.array([hashlib.sha256(x.tobytes()).hexdigest() for x in truths])
    for v in range(2):
        b = views[v]
        r = orc.Rasterizer(np.float32)
        img, R = r.forward(1, M, b[37:40], W, H, s["loc"], s["sh"], s["opac"], s["scale"], 1.0, s["rot"], b[0:16], b[16:32], b[32:35],
                           float(b[35]), float(b[36]))
        out[f"v{v}_num_rendered"] = R
        out[f"v{v}_point_list_sha256"] = hashlib.sha256(r.get("point_list").tobytes()).hexdigest()
        out[f"v{v}_ranges_sha256"] = hashlib.sha256(r.get("ranges").tobytes()).hexdigest()
        out[f"v{v}_radii_sha256"] = hashlib.sha256(r.get("radii").tobytes()).hexdigest()
        out[f"v{v}_point_list_head"] = r.get("point_list")[:256]
        out[f"v{v}_image_crop"] = img[:, 96:160, 96:160].copy()
        out[f"v{v}_image_mean"] = img.reshape(3, -1).mean(1)
        out[f"v{v}_final_T_crop"] = r.get("final_T").reshape(H, W)[96:160, 96:160].copy()
        out[f"v{v}_n_contrib_sum"] = int(r.get("n_contrib").astype(np.int64).sum())
        out[f"v{v}_margin_crop"] = r.get("margin").reshape(H, W)[96:160, 96:160].copy()
    o = orc.train_views(P, 1, M, W, H, s["loc"], s["sh"], s["scale"], s["opac"], s["rot"], views, np.concatenate(truths), 2.0)
    for k in ("var", "loc", "sh", "scale", "opac", "rot"):
        out["avg_" + k] = o[k]
    return out, truths


# The seam-level parity cases of tests/test_gpu_raster.py::CASES beyond cfg1 (ragged sizes, SH degree 0 / 2 / 3, 79 splat
# blocks): fingerprints of the oracle's forward AND backward, small enough to commit, so that an edit of the oracle cannot
# drift silently on the cases the GPU path is compared with.
SEAM_CASES = [
    # P,   M, D, W,   H,   seed
    (800, 1, 0, 250, 130, 11),
    (600, 9, 2, 96, 160, 12),
    (500, 16, 3, 128, 128, 13),
    (20000, 1, 0, 200, 72, 14),
]


def build_seam_cases():
    out = {}
    for n, (P, M, D, W, H, seed) in enumerate(SEAM_CASES):
        s = gs.synth.random_splats(P, M, seed)
        views = gs.camera.train_views(gs.camera.get_cameras(2), W, H)
        b = views[1]                                   # the pass test_backward_parity takes: camera 1, white background
        r = orc.Rasterizer(np.float32)
        img, R = r.forward(D, M, b[37:40], W, H, s["loc"], s["sh"], s["opac"], s["scale"], 1.0, s["rot"], b[0:16], b[16:32], b[32:35],
                           float(b[35]), float(b[36]))
        dpix = np.random.default_rng(seed).uniform(-1, 1, (3, H, W)).astype(np.float32)
        g = r.backward(dpix)
        k = f"c{n}_"
        out[k + "shape"] = np.array([P, M, D, W, H, seed], np.int64)
        out[k + "num_rendered"] = R
        for name in ("point_list", "ranges", "radii", "tiles_touched", "n_contrib"):
            out[k + name + "_sha256"] = hashlib.sha256(r.get(name).tobytes()).hexdigest()
        out[k + "image_mean"] = img.reshape(3, -1).mean(1)
        out[k + "image_row"] = img[:, H // 2, :].copy()
        out[k + "final_T_row"] = r.get("final_T").reshape(H, W)[H // 2].copy()
        for name, stride in (("dL_dmean3D", 3), ("dL_dsh", 3 * M), ("dL_dscale", 3), ("dL_drot", 4), ("dL_dopacity", 1), ("dL_dcov3D", 6)):
            a = g[name].reshape(P, stride)
            out[k + name + "_head"] = a[:64].copy()                      # the first 64 splats, every component
            out[k + name + "_abs_sum"] = np.abs(a.astype(np.float64)).sum(0)   # and the whole array's weight per component
    return out


# The accounting quantities of the parity budget on ONE scene of the seeded sweep (tests/test_gpu_sweep.py, scene 12: 300 needle-prone
# splats, 4 passes @184x144): sum|term|, the decision-flip allowance, the conditioning bound and the measured chain noise are what a
# GPU result is allowed to deviate by — an edit of the oracle that moves THEM would loosen or tighten every sweep assertion silently.
def build_budget_case():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_sweep import KAPPA, wild_rig, wild_scene
    from util import step_budget
    rng = np.random.default_rng(0x5EED5EED + 12)
    s, kind = wild_scene(rng)
    wild_scene(rng)
    P, M = s["count"], s["M"]
    W, H = int(rng.integers(17, 210)), int(rng.integers(17, 210))
    cams = wild_rig(rng)
    views = gs.camera.train_views(cams, W, H)
    b = views[0]
    r = orc.Rasterizer(np.float32)
    img, R = r.forward(s["D"], M, b[37:40], W, H, s["loc"], s["sh"], s["opac"], s["scale"], 1.0, s["rot"], b[0:16], b[16:32], b[32:35], float(b[35]), float(b[36]))
    dpix = np.random.default_rng(12).uniform(-1, 1, (3, H, W)).astype(np.float32)
    g = r.backward(dpix, want_abs=True, flip_margin=1e-3, want_cond=True, power_ulps=KAPPA)
    out = dict(shape=np.array([P, M, W, H, len(cams), R], np.int64), kind=kind)
    for name in ("abs9", "flip9", "cond9"):
        out[name + "_per_sum"] = g[name].sum(0)
        out[name + "_head"] = g[name][:32].copy()
    out["margin_min"] = float(r.get("margin").min())
    truths = np.zeros((len(views), W * H), np.uint32)
    bud = step_budget(orc, s, s["D"], M, W, H, views, truths, float(len(views)), flip_margin=1e-3, chain_noise_trials=8, cond_kappa=KAPPA)
    for k in ("loc", "sh", "scale", "opac", "rot", "var"):
        out["budget_" + k + "_sum"] = float(bud[k]["budget"].sum())
        out["want_" + k + "_abs_sum"] = float(np.abs(bud[k]["want"].astype(np.float64)).sum())
    out["noise_scale_sum"] = float(bud["scale"]["noise"].sum())
    return out


def build_reference_noise_case():
    """The reference's own run-to-run envelope (oracle atomic_prepare / atomic_sums) on BASELINE cfg1's step: per gradient array the envelope
    [min, max] over eight seeded fp32 atomicAdd orders, the double-summed value and the four order families' sums of the first pass —
    the yardstick of tests/test_gpu_envelope.py must not move when the oracle is edited."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import oracle_forward, step_budget, view_parts
    P, M, V, W, H = gs.synth.CONFIGS[1]
    seed = gs.synth.seed_for(1)
    s = gs.synth.random_splats(P, M, seed)
    cams = gs.camera.get_cameras(1)
    views = gs.camera.train_views(cams, W, H)
    t = gs.synth.random_splats(P // 2, M, seed + 1000)
    truth = []
    for v in range(2):
        r, img, _ = oracle_forward(orc, t, t["D"], M, view_parts(views[v]), W, H)
        truth.append(orc.image_float_to_int(img, W, H))
    bud = step_budget(orc, s, s["D"], M, W, H, views, np.concatenate(truth), 2.0, atomic_seeds=range(8))
    out = {}
    for k in ("loc", "sh", "scale", "opac", "rot", "var"):
        runs = bud["runs"][k]
        out[k + "_lo"], out[k + "_hi"], out[k + "_want"] = runs.min(0), runs.max(0), bud[k]["want"]
    r, img, _ = oracle_forward(orc, s, s["D"], M, view_parts(views[0]), W, H)
    dpix = orc.image_int_to_loss(truth[0], img, W, H)
    r.atomic_prepare(dpix)
    for mode in range(4):
        out[f"sums9_mode{mode}"] = r.atomic_sums(3, mode)
    r.atomic_release()
    return out


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    out, _ = build()
    path = os.path.join(here, "cfg1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
    path = os.path.join(here, "seam_cases.npz")
    np.savez_compressed(path, **build_seam_cases())
    print("wrote", path, os.path.getsize(path), "bytes")
    path = os.path.join(here, "budget_case.npz")
    np.savez_compressed(path, **build_budget_case())
    print("wrote", path, os.path.getsize(path), "bytes")
    path = os.path.join(here, "reference_noise_case.npz")
    np.savez_compressed(path, **build_reference_noise_case())
    print("wrote", path, os.path.getsize(path), "bytes")
