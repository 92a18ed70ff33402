"""Diagnostic: the long-list seam scene of tests/test_gpu_raster.py, printing the out-of-budget splats' sums."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import util
from util import view_parts, SeamRaster, oracle_forward
from oracle import pyoracle as orc
from test_gpu_raster import NINE

P = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
M, D, W, H = 1, 0, 32, 32
s = util.gs.synth.random_splats(P, M, 99)
s["loc"] = (s["loc"] * 0.05).astype(np.float32)
s["opac"] = (s["opac"] * 0.02).astype(np.float32)
cams = util.gs.camera.get_cameras(1, 10.0, 20.0)
vp = view_parts(util.gs.camera.train_views(cams, W, H)[0])
sr = SeamRaster()
out, R = sr.forward(s, D, M, vp, W, H)
r, oout, oR = oracle_forward(orc, s, D, M, vp, W, H)
dpix = np.ones((3, H, W), np.float32)
g = sr.backward(dpix)
og = r.backward(dpix, want_abs=True, flip_margin=1e-3)
abs9, flip9 = og["abs9"], og["flip9"]
for name, (qs, stride, cols) in NINE.items():
    got = g[name].reshape(P, stride).astype(np.float64); want = og[name].reshape(P, stride).astype(np.float64)
    for q, c in zip(qs, cols):
        err = np.abs(got[:, c] - want[:, c])
        rel = err / np.maximum(abs9[:, q], 1e-30)
        tol = 1e-4 * np.maximum(abs9[:, q], 1e-3 * abs9[:, q].max() + 1e-30) + flip9[:, q]
        bad = np.flatnonzero(err > tol)
        print(f"q={q} {name}[{c}]: err/sum|term| median {np.median(rel):.2e} p99 {np.quantile(rel, 0.99):.2e} max {rel.max():.2e}; bad {bad.size}", bad[:6],
              [(f"{got[i, c]:.6g}", f"{want[i, c]:.6g}", f"{abs9[i, q]:.3g}", f"{flip9[i, q]:.3g}") for i in bad[:3]])
