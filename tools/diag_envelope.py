#!/usr/bin/env python3
"""How far the HIP step's averaged gradients lie from the reference's own run-to-run envelope (K seeded fp32 atomicAdd
orders, oracle/gs_oracle.cpp atomic_prepare / atomic_sums) — the measurement behind tests/test_gpu_envelope.py.
usage (GPU box): python tools/diag_envelope.py [cfg ...]   cfg in 1 2 3 s1 s2"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gsplat_amd as gs            # noqa: E402
from oracle import pyoracle as orc  # noqa: E402
import util                         # noqa: E402
import test_gpu_trainer as tg       # noqa: E402

CFG = {"1": (1000, 4, 1, 256, 256), "s1": (1500, 1, 3, 128, 96), "s2": (700, 16, 2, 112, 112), "2": (10000, 1, 4, 512, 512),
       "3": (100000, 16, 8, 1024, 1024)}
K = 8
FS = (1, 2, 4, 8, 16, 64, 256)


def rates(got, lo, hi, want):
    got = np.asarray(got, np.float64)
    mid, hw = (lo + hi) / 2, (hi - lo) / 2
    d = util.envelope_distance(got, lo, hi)
    strict = np.abs(got - want) <= 1e-4 * np.abs(want)
    out = {"strict_1e-4_of_value": float(strict.mean()), "inside": float((d == 0).mean())}
    for F in FS:
        out[f"within_{F}x"] = float((np.abs(got - mid) <= F * hw).mean())
    return out, d, strict


def main():
    which = sys.argv[1:] or ["1", "s1", "s2", "2"]
    res = {}
    for c in which:
        P, M, n_cams, W, H = CFG[c]
        s, cams, fw, fb, tr = tg._setup(orc, P, M, n_cams, W, H, 0x5EED0001)
        views = gs.camera.train_views(cams, W, H)
        truths = np.concatenate(fw + fb)
        tr.accumulate(stats=True)
        g = tg._read_grads(tr, P, M)
        proj = gs.Project()
        tr.apply(proj)
        host = gs.ModelSplatsHost.fromVectors(s["loc"], s["sh"], s["scale"], s["opac"], s["rot"])
        host.shDegree = s["D"]
        tr.model = gs.ModelSplatsDevice(host)
        tr.train(proj, densify=False, stats=True)
        gf = tg._read_grads(tr, P, M)
        tr.close()
        t0 = time.time()
        bud = util.step_budget(orc, s, s["D"], M, W, H, views, truths, 2.0 * n_cams, atomic_seeds=range(K))
        print(f"[cfg {c}: {P} splats, {2 * n_cams} passes @{W}x{H}] oracle + budget + {K} reference runs in {time.time() - t0:.1f} s", flush=True)
        res[c] = {}
        for k in ["loc", "sh", "scale", "opac", "rot", "var"]:
            runs = bud["runs"][k].astype(np.float64)
            want = bud[k]["want"].astype(np.float64)
            lo, hi = util.envelope(runs)
            hw = (hi - lo) / 2
            row = {}
            for form, got in (("per_pass", g[k]), ("fused", gf[k])):
                if form == "fused" and k == "var":
                    continue
                r, d, strict = rates(got, lo, hi, want)
                fl = bud[k]["flip"] > 0
                # entries neither within 1e-4 of the value nor within the 16x widened envelope
                far = ~strict & (np.abs(np.asarray(got, np.float64) - (lo + hi) / 2) > 16 * hw)
                r["far_total"] = int(far.sum()); r["far_with_flip"] = int((far & fl).sum())
                worst = np.abs(np.asarray(got, np.float64) - (lo + hi) / 2)[far & ~fl] / np.maximum(hw[far & ~fl], 1e-300) if (far & ~fl).any() else np.zeros(0)
                r["far_noflip_worst_x"] = float(worst.max()) if worst.size else 0.0
                # same entries: error in units of 2^-24 * sum|term| carried (sumabs)
                sa = bud[k]["sumabs"]
                e = np.abs(np.asarray(got, np.float64) - want)
                m = far & ~fl & (sa > 0)
                r["far_noflip_worst_ulps_of_sumabs"] = float((e[m] / (2.0 ** -24 * sa[m])).max()) if m.any() else 0.0
                row[form] = r
            # the yardstick: each reference run against the envelope of the OTHER K - 1
            loo = []
            for j in range(K):
                others = np.delete(runs, j, axis=0)
                lo2, hi2 = others.min(0), others.max(0)
                r, _, _ = rates(runs[j], lo2, hi2, want)
                loo.append(r)
            row["ref_leave_one_out"] = {a: float(np.mean([x[a] for x in loo])) for a in loo[0]}
            row["entries"] = int(want.size)
            row["median_budget_over_halfwidth"] = float(np.median((bud[k]["budget"][hw > 0]) / hw[hw > 0])) if (hw > 0).any() else None
            res[c][k] = row
            print(k, json.dumps(row), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "r5_envelope_diag.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
