#!/usr/bin/env python3
"""Useful-lane fraction of a "hit" in the two blend loops.  Needs a library built with
-DGS_DIAG_COUNT_ACTIVE:
    tools/build_variant.sh lanes k_render.hip -DGS_DIAG_COUNT_ACTIVE
    GSPLAT_MI355_LIB=$PWD/tools/_variants/lib_lanes.so python tools/diag_lanes.py --config 3 [--views 16] [--steps 3]
A hit = one evaluated (tile entry, 8x8 pixel block) pair, 64 lanes wide and branch-free; a lane is useful when its pixel is still
blending and the pair passes the alpha >= 1/255 test there.  Prints one JSON line."""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import gsplat_amd as gs  # noqa: E402
from gsplat_amd import capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=3)
ap.add_argument("--views", type=int, default=0)
ap.add_argument("--steps", type=int, default=3)
args = ap.parse_args()
L = capi.lib()
P, M, V, W, H = gs.synth.CONFIGS[args.config]
V = args.views or V
D = gs.synth.sh_degree_for(M)
n_cams = max(V // 2, 1)
s = gs.synth.random_splats(P, M, gs.synth.seed_for(args.config))
t = gs.synth.random_splats(max(P // 2, 1), M, gs.synth.seed_for(args.config) + 1000)
cams = gs.camera.get_cameras(n_cams)
tr = gs.Trainer(W, H)
th = gs.ModelSplatsHost.fromVectors(t["loc"], t["sh"], t["scale"], t["opac"], t["rot"]); th.shDegree = D
tr.model = gs.ModelSplatsDevice(th)
fw = [tr.render(W, H, 1.0, c, background=(1.0, 1.0, 1.0)).reshape(-1) for c in cams]
fb = [tr.render(W, H, 1.0, c, background=(0.0, 0.0, 0.0)).reshape(-1) for c in cams]
h = gs.ModelSplatsHost.fromVectors(s["loc"], s["sh"], s["scale"], s["opac"], s["rot"]); h.shDegree = D
tr.model = gs.ModelSplatsDevice(h)
tr.captureTruths(cams, fw, fb)
proj = gs.Project(updateRule=capi.GS_UPDATE_ADAM)
st = tr.train(proj, stats=True)   # warm-up (arena growth replays would count twice)
cnt = (C.c_ulonglong * 8)()
capi.check(L.gs_debug_counters(cnt, 1))
for _ in range(args.steps):
    st = tr.train(proj, stats=True)
capi.check(L.gs_debug_counters(cnt, 0))
c = [int(x) for x in cnt]
if c[0] == 0:
    raise SystemExit("counters are zero: the library was not built with -DGS_DIAG_COUNT_ACTIVE (see the docstring)")
out = {"config": args.config, "splats": P, "views": 2 * n_cams, "width": W, "height": H, "steps": args.steps,
       "entries_per_camera": st.num_rendered / st.views, "max_tile_list": st.max_tile_list,
       "forward": {"staged_pairs": c[4], "hits": c[0], "active_lanes": c[1], "hit_fraction_of_staged": c[0] / max(c[4], 1),
                   "useful_lane_fraction_of_a_hit": c[1] / (64.0 * c[0])},
       "backward": {"staged_pairs": c[5], "hits": c[2], "active_lanes": c[3], "hit_fraction_of_staged": c[2] / max(c[5], 1),
                    "useful_lane_fraction_of_a_hit": c[3] / (64.0 * max(c[2], 1)),
                    "iterations_if_packed_by_8x4_half": c[6], "iterations_if_packed_by_4x4_quadrant": c[7],
                    "packing_ceiling": {"8x4": c[6] / max(c[2], 1), "4x4": c[7] / max(c[2], 1),
                                        "note": "iterations relative to today's hits if two / four entries shared a wave iteration (per 64-entry round: the largest "
                                                "per-half / per-quadrant hit count); every packed iteration would read its entries' records with per-lane LDS addresses"}},
       "note": "hit = evaluated (tile entry, 8x8 block) pair; staged = pairs in front of the exact block test; useful lane = pixel alive and alpha >= 1/255"}
print(json.dumps(out))
tr.close()
