"""Diagnostic: the step-level parity accounting of tests/test_gpu_trainer.py for one configuration, printing the
unexplained entries.   python tools/diag_step.py P M n_cams W H"""
import sys, os
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests")); sys.path.insert(0, os.path.join(HERE, ".."))
import numpy as np
import gsplat_amd as gs
from oracle import pyoracle as orc
from util import step_budget
from test_gpu_trainer import _setup, _read_grads

P, M, n_cams, W, H = (int(x) for x in sys.argv[1:6])
s, cams, fw, fb, tr = _setup(orc, P, M, n_cams, W, H, 0x5EED0001)
views = gs.camera.train_views(cams, W, H)
truths = np.concatenate(fw + fb)
o = orc.train_views(P, s["D"], M, W, H, s["loc"], s["sh"], s["scale"], s["opac"], s["rot"], views, truths, 2.0 * n_cams)
st = tr.accumulate(stats=True)
g = _read_grads(tr, P, M)
margin = 1e-4 if st.max_tile_list <= 1024 else 1e-3
bud = step_budget(orc, s, s["D"], M, W, H, views, truths, 2.0 * n_cams, flip_margin=margin)
stride = dict(loc=3, sh=3 * M, scale=3, opac=1, rot=4, var=1)
for k in ["loc", "sh", "scale", "opac", "rot", "var"]:
    err = np.abs(g[k].astype(np.float64) - o[k]); tol = bud[k]["budget"] + 4e-6 * np.abs(o[k]) + 1e-37
    bad = np.flatnonzero(err > tol)
    print(k, "unexplained", bad.size, "worst", (err / tol).max())
    for i in bad[:4]:
        sp = i // stride[k]
        print("   entry", i, "splat", sp, "got", g[k][i], "want", o[k][i], "budget", bud[k]["budget"][i], "sumabs", bud[k]["sumabs"][i],
              "| scale", s["scale"][3 * sp:3 * sp + 3], "|rot|", np.linalg.norm(s["rot"][4 * sp:4 * sp + 4]), "opac", s["opac"][sp],
              "| loc err/tol of the splat", (np.abs(g["loc"].astype(np.float64) - o["loc"]) / (bud["loc"]["budget"] + 1e-37))[3 * sp:3 * sp + 3],
              "| scale got/want of the splat", g["scale"][3 * sp:3 * sp + 3], o["scale"][3 * sp:3 * sp + 3])
