#!/usr/bin/env python3
"""Second overlap probe: TWO trainers in ONE process, each on its own HIP stream and host thread, each stepping cfg3 on half of
the cameras — against one trainer on all of them.  tools/overlap_probe.sh did this with two processes (+4.8 %); queues of one
process share the device more finely than two processes do, so this is the tighter ceiling for camera-group pipelining
inside one trainer (which would still share the projection and join for the reduce and the update; the two trainers here
each run their own update, i.e. one more 0.037 ms launch per 16 views than a pipelined trainer would).
    gpurun -- 'python tools/overlap_probe2.py [--steps 600]'
"""
import argparse
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import gsplat_amd as gs  # noqa: E402
from gsplat_amd import capi  # noqa: E402


def make(cams, framesW, framesB, host, W, H):
    tr = gs.Trainer(W, H)
    tr.model = gs.ModelSplatsDevice(host)
    tr.captureTruths(cams, framesW, framesB)
    return tr


def run(trainers, steps, proj):
    """steps of every trainer, one host thread each; wall time from a common start to the last synchronize"""
    for tr in trainers:
        for _ in range(20):
            tr.train(proj, densify=False)
        tr.synchronize()
    go = threading.Barrier(len(trainers) + 1)
    done = []

    def work(tr):
        go.wait()
        for _ in range(steps):
            tr.train(proj, densify=False)
        tr.synchronize()
        done.append(time.perf_counter())
    th = [threading.Thread(target=work, args=(tr,)) for tr in trainers]
    for t in th:
        t.start()
    go.wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    return max(done) - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--groups", type=int, default=2)
    a = ap.parse_args()
    P, M, V, W, H = gs.synth.CONFIGS[a.config]
    D = gs.synth.sh_degree_for(M)
    n = V // 2
    seed = gs.synth.seed_for(a.config)
    s = gs.synth.random_splats(P, M, seed)
    t = gs.synth.random_splats(max(P // 2, 1), M, seed + 1000)
    cams = gs.camera.get_cameras(n)
    thost = gs.ModelSplatsHost.fromVectors(t["loc"], t["sh"], t["scale"], t["opac"], t["rot"])
    thost.shDegree = D
    r = gs.Trainer(W, H)
    r.model = gs.ModelSplatsDevice(thost)
    fW = [r.render(W, H, 1.0, c, background=(1.0, 1.0, 1.0)).reshape(-1) for c in cams]
    fB = [r.render(W, H, 1.0, c, background=(0.0, 0.0, 0.0)).reshape(-1) for c in cams]
    r.close()
    host = gs.ModelSplatsHost.fromVectors(s["loc"], s["sh"], s["scale"], s["opac"], s["rot"])
    host.shDegree = D
    # learning rates 0: the workload stays the metric's over any number of steps ("what a long run measures")
    proj = gs.Project(updateRule=capi.GS_UPDATE_ADAM, lrLocation=0.0, lrSh=0.0, lrScale=0.0, lrOpacity=0.0, lrRotation=0.0)
    out = {"config": a.config, "steps": a.steps, "cameras": n, "groups": a.groups}
    whole = make(cams, fW, fB, host, W, H)
    dt = run([whole], a.steps, proj)
    out["one_trainer_all_cameras"] = {"ms_per_step": dt / a.steps * 1e3, "views_per_s": V * a.steps / dt}
    whole.close()
    k = n // a.groups
    parts = [make(cams[g * k:(g + 1) * k], fW[g * k:(g + 1) * k], fB[g * k:(g + 1) * k], host, W, H) for g in range(a.groups)]
    dt = run(parts[:1], a.steps, proj)
    out["one_trainer_one_group"] = {"ms_per_step": dt / a.steps * 1e3, "views_per_s": 2 * k * a.steps / dt}
    dt = run(parts, a.steps, proj)
    out["all_groups_side_by_side"] = {"ms_per_step_of_each": dt / a.steps * 1e3, "views_per_s": 2 * k * a.groups * a.steps / dt}
    out["side_by_side_over_one_trainer"] = out["all_groups_side_by_side"]["views_per_s"] / out["one_trainer_all_cameras"]["views_per_s"]
    for p in parts:
        p.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
