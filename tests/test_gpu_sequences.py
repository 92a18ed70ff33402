"""Call sequences a GUI host produces and a benchmark does not (src/ui/UiFrame.cpp, UiPanelViewOutput.cpp, UiPanelToolsTruth.cpp drive
one Trainer for the life of the window): preview renders between the halves of an iteration, truth sets and models swapped under a
live trainer, empty and full models, update rules switched between iterations, trainers created and destroyed by the dozen.
Everything here is checked against the same calls in the plain order, bit for bit — no oracle involved."""
import ctypes as C

import numpy as np
import pytest

import gsplat_amd as gs
from gsplat_amd import capi

pytestmark = pytest.mark.gpu


def _scene(P=900, M=4, n_cams=2, W=80, H=64, seed=77):
    s = gs.synth.random_splats(P, M, seed)
    cams = gs.camera.get_cameras(n_cams)
    rng = np.random.default_rng(seed)
    fw = [rng.integers(0, 2 ** 32, W * H, dtype=np.uint32) | np.uint32(0xFF000000) for _ in cams]
    fb = [rng.integers(0, 2 ** 32, W * H, dtype=np.uint32) | np.uint32(0xFF000000) for _ in cams]
    host = gs.ModelSplatsHost.fromVectors(s["loc"], s["sh"], s["scale"], s["opac"], s["rot"])
    host.shDegree = s["D"]
    return s, cams, fw, fb, host, W, H


def _bits(tr):
    h = gs.ModelSplatsHost.fromDevice(tr.model)
    n, M = h.count, h.shCoeffs
    return n, np.concatenate([h.locations[:3 * n], h.shs[:3 * M * n], h.scales[:3 * n], h.opacities[:n], h.rotations[:4 * n]]).view(np.uint32).copy()


def test_preview_render_between_accumulate_and_apply_changes_nothing():
    """The preview panel renders on every idle event (src/ui/UiPanelViewOutput.cpp:52-60), i.e. also between a host's
    gs_trainer_accumulate and gs_trainer_apply: the render shares the trainer's binning scratch and must leave the averaged
    gradients, the optimizer state and the model alone."""
    s, cams, fw, fb, host, W, H = _scene()
    out = []
    for with_render in (False, True):
        tr = gs.Trainer(W, H)
        tr.model = gs.ModelSplatsDevice(host)
        tr.captureTruths(cams, fw, fb)
        proj = gs.Project(updateRule=capi.GS_UPDATE_ADAM)
        frames = []
        for it in range(4):
            tr.accumulate()
            if with_render:
                frames.append(tr.render(50 + 13 * it, 40 + 7 * it, 1.0 + 0.5 * it, cams[it % 2]))
            tr.apply(proj, densify=(it == 2))
            if with_render:
                frames.append(tr.render(W, H, 1.0, cams[0]))
        out.append(_bits(tr))
        assert not with_render or all(f.any() for f in frames)
        tr.close()
    assert out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1])


def test_truths_and_models_swapped_under_a_live_trainer():
    """captureTruths with another camera count (UiPanelToolsTruth::onButtonCapture after the sphere counts changed) and a new model
    with another SH size (File > Load) on the same Trainer: the next iteration equals that of a fresh trainer given the same state."""
    s, cams, fw, fb, host, W, H = _scene()
    s2, cams3, fw3, fb3, host2, _, _ = _scene(P=1500, M=16, n_cams=3, seed=78)
    proj = gs.Project()
    live = gs.Trainer(W, H)
    live.model = gs.ModelSplatsDevice(host)
    live.captureTruths(cams, fw, fb)
    for _ in range(2):
        live.train(proj, densify=False)
    live.captureTruths(cams3, fw3, fb3)            # more cameras, same model
    live.train(proj, densify=False)
    n_mid, mid = _bits(live)
    live.model = gs.ModelSplatsDevice(host2)       # bigger model, 16 SH coefficients instead of 4
    live.train(proj, densify=True)
    live.captureTruths(cams[:1], fw[:1], fb[:1])   # fewer cameras
    st = live.train(proj, densify=False, stats=True)
    assert st.views == 2
    n_live, bits_live = _bits(live)
    # the same history on fresh trainers
    a = gs.Trainer(W, H)
    a.model = gs.ModelSplatsDevice(host)
    a.captureTruths(cams, fw, fb)
    for _ in range(2):
        a.train(proj, densify=False)
    b = gs.Trainer(W, H)
    b.model = gs.ModelSplatsDevice(gs.ModelSplatsHost.fromDevice(a.model))
    b.captureTruths(cams3, fw3, fb3)
    b.train(proj, densify=False)
    n_b, bits_b = _bits(b)
    assert n_b == n_mid and np.array_equal(bits_b, mid)
    c = gs.Trainer(W, H)
    c.model = gs.ModelSplatsDevice(host2)
    c.captureTruths(cams3, fw3, fb3)
    c.train(proj, densify=True)
    d = gs.Trainer(W, H)
    d.model = gs.ModelSplatsDevice(gs.ModelSplatsHost.fromDevice(c.model))
    d.captureTruths(cams[:1], fw[:1], fb[:1])
    d.train(proj, densify=False)
    n_d, bits_d = _bits(d)
    assert n_d == n_live and np.array_equal(bits_d, bits_live)
    for t in (live, a, b, c, d):
        t.close()


def test_empty_and_full_models_step_and_densify():
    """count = 0 (File > New before any field is initialised) and count = capacity (no room to split or clone): train with and
    without densify runs, counts stay inside [0, capacity], and an emptied model keeps stepping."""
    s, cams, fw, fb, host, W, H = _scene(P=300)
    L = capi.lib()
    tr = gs.Trainer(W, H)
    empty = gs.ModelSplatsHost(64, 1, 4)
    empty.count = 0
    tr.model = gs.ModelSplatsDevice(empty)
    tr.captureTruths(cams, fw, fb)
    proj = gs.Project(paramDensifyVariance=1e-9, paramSplitSize=1e-3, paramCullOpacity=0.0)   # everything wants to split
    for densify in (False, True, False):
        st = tr.train(proj, densify=densify, stats=True)
        assert st.count_before == 0 and st.count_after == 0 and st.num_rendered == 0
    full = gs.ModelSplatsHost(300, 1, 4)         # capacity = count: the five-vector constructor would reserve a million
    full.count = 300
    full.locations[:], full.shs[:], full.scales[:], full.opacities[:], full.rotations[:] = s["loc"], s["sh"], s["scale"], s["opac"], s["rot"]
    tr.model = gs.ModelSplatsDevice(full)
    assert tr.model.capacity == 300
    st = tr.train(proj, densify=True, stats=True)
    assert st.count_before == 300 and 0 <= st.count_after <= 300
    prune_all = gs.Project(paramCullOpacity=2.0)       # every opacity is below 2: the model empties
    st = tr.train(prune_all, densify=True, stats=True)
    assert st.count_after == 0
    st = tr.train(proj, densify=True, stats=True)
    assert st.count_before == 0 and st.count_after == 0
    fbuf = tr.render(33, 21, 1.0, cams[0], background=(1.0, 0.0, 0.0))
    assert np.all(fbuf == fbuf.flat[0])                # an empty model renders the background
    tr.close()


def test_update_rule_switched_between_iterations():
    """gs_hyper.update_rule is read per call: SGD and Adam iterations interleave; the Adam moments sleep through the SGD iterations and
    the result equals the same sequence run on a trainer restored from a checkpoint taken in the middle."""
    s, cams, fw, fb, host, W, H = _scene()
    sgd, adam = gs.Project(), gs.Project(updateRule=capi.GS_UPDATE_ADAM)
    order = [sgd, adam, adam, sgd, adam, sgd, sgd, adam]
    tr = gs.Trainer(W, H)
    tr.model = gs.ModelSplatsDevice(host)
    tr.captureTruths(cams, fw, fb)
    for p in order[:4]:
        tr.train(p, densify=False)
    saved = gs.ModelSplatsHost.fromDevice(tr.model)
    m1, m2, steps = tr.adam_state()
    assert steps == 2
    for p in order[4:]:
        tr.train(p, densify=False)
    n1, want = _bits(tr)
    re = gs.Trainer(W, H)
    re.model = gs.ModelSplatsDevice(saved)
    re.captureTruths(cams, fw, fb)
    re.set_adam_state(m1, m2, steps)
    for p in order[4:]:
        re.train(p, densify=False)
    n2, got = _bits(re)
    assert n1 == n2 and np.array_equal(got, want)
    tr.close(); re.close()


def test_trainers_and_models_by_the_dozen_leave_no_memory_behind():
    """Forty trainers (each with scratch for its image size, a model, truth images, a few iterations with densify, a render) created
    and destroyed one after the other: the device's free memory ends where it began (within the allocator's slack)."""
    hip = C.CDLL("libamdhip64.so")               # the runtime the library itself runs on (already loaded into this process)

    def free_bytes():
        free, total = C.c_size_t(), C.c_size_t()
        assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
        return free.value
    s, cams, fw, fb, host, W, H = _scene()

    def cycle(k):
        tr = gs.Trainer(W + 16 * (k % 3), H + 16 * (k % 2))
        rng = np.random.default_rng(k)
        n = tr.width * tr.height
        f1 = [rng.integers(0, 2 ** 32, n, dtype=np.uint32) for _ in cams]
        tr.model = gs.ModelSplatsDevice(host)
        tr.captureTruths(cams, f1, f1)
        proj = gs.Project(updateRule=capi.GS_UPDATE_ADAM if k % 2 else capi.GS_UPDATE_SGD_CLAMP)
        for it in range(3):
            tr.train(proj, densify=(it == 1))
        tr.render(40, 30, 1.0, cams[0])
        clone = gs.ModelSplatsDevice(gs.ModelSplatsHost.fromDevice(tr.model))
        tr.synchronize()
        tr.close()
        del clone
    for k in range(3):           # warm-up: first-use allocations of the runtime itself
        cycle(k)
    capi.check(capi.lib().gs_device_synchronize())
    free0 = free_bytes()
    for k in range(40):
        cycle(k)
    capi.check(capi.lib().gs_device_synchronize())
    free1 = free_bytes()
    assert free0 - free1 < 64 << 20, (free0, free1)
