#!/usr/bin/env python3
"""Soak run of the driver loop on the GPU: cfg2-sized model, truths from a hidden target model rendered by the
product's preview path, re-captured with re-rotated camera spheres every 40 iterations, densify/prune every 50,
Adam.  Prints loss, splat count and device memory along the way; fails on a non-finite model or a growing footprint.
  python tools/soak.py [iterations]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
import gsplat_amd as gs
from gsplat_amd import capi
import ctypes as C

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
P, M, W, H = 10000, 4, 256, 256
target = gs.synth.random_splats(6000, M, 1)
thost = gs.ModelSplatsHost.fromVectors(target["loc"], target["sh"], target["scale"], target["opac"], target["rot"]); thost.shDegree = 1
renderer = gs.Trainer(W, H); renderer.model = gs.ModelSplatsDevice(thost)

def capture(cameras):
    fw = [renderer.render(W, H, 1.0, c, background=(1.0, 1.0, 1.0)).reshape(-1) for c in cameras]
    fb = [renderer.render(W, H, 1.0, c, background=(0.0, 0.0, 0.0)).reshape(-1) for c in cameras]
    return fw, fb

s = gs.synth.random_splats(P, M, 2)
host = gs.ModelSplatsHost.fromVectors(s["loc"], s["sh"], s["scale"], s["opac"], s["rot"]); host.shDegree = 1
tr = gs.Trainer(W, H); tr.model = gs.ModelSplatsDevice(host)
proj = gs.Project(updateRule=capi.GS_UPDATE_ADAM, lrLocation=2e-3, lrSh=5e-3, lrScale=5e-4, lrOpacity=5e-3, lrRotation=1e-3)
proj.intervalCapture, proj.intervalDensify = 40, 50
auto = gs.driver.AutoTrainer(tr, proj, capture)

def mem_free():
    free, total = C.c_size_t(), C.c_size_t()
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemGetInfo(C.byref(free), C.byref(total))
    return free.value

t0 = time.time(); free0 = None; losses = []
for it in range(iters):
    auto.step()
    if it % 100 == 99:
        st = tr.train(proj, stats=True)
        h = gs.ModelSplatsHost.fromDevice(tr.model)
        ok = all(np.isfinite(a[:k * h.count]).all() for a, k in ((h.locations, 3), (h.shs, 3 * M), (h.scales, 3), (h.opacities, 1), (h.rotations, 4)))
        f = mem_free()
        free0 = free0 or f
        losses.append(st.loss)
        print(f"iter {proj.iterations:5d}  splats {h.count:7d}  loss {st.loss:10.1f}  device free {f / 2**30:7.2f} GiB  finite {ok}  {time.time() - t0:6.1f} s", flush=True)
        assert ok, "non-finite parameters"
        assert f > free0 - (2 << 30), "device memory footprint keeps growing"
print("soak ok", "loss first/last", losses[0], losses[-1])
