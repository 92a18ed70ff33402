"""CPU tests of the rows either side of the training path: field initialisers (N3,
src/ui/UiFrame.cpp:137-264), the .gobj / settings.json formats (N2, :323-450), the two-sphere camera
rig (src/Camera.cpp:33-58) and the auto-train driver loop (N4, :266-298)."""
import json
import math

import numpy as np
import pytest

import gsplat_amd as gs


def test_field_grid_and_mono():
    h = gs.fields.initFieldGrid()
    assert (h.count, h.capacity, h.shDegree, h.shCoeffs) == (17 ** 3, 1000000, 1, 4)
    loc = h.locations[:3 * h.count].reshape(-1, 3)
    assert loc.min() == -4 and loc.max() == 4 and np.array_equal(loc[1], [-4, -4, -3.5])       # z fastest, step 0.5
    assert np.all(h.scales[:3 * h.count] == np.float32(0.5) * np.float32(0.1)) and np.all(h.opacities[:h.count] == 1)
    assert not h.shs[:12 * h.count].any()
    assert np.array_equal(h.rotations[:4], [0, 0, 0, 1])                                        # glm::quat memory {x,y,z,w}
    assert np.array_equal(gs.fields.initFieldGrid(quat_xyzw=False).rotations[:4], [1, 0, 0, 0])
    m = gs.fields.initFieldMono()
    assert m.count == 1 and np.allclose(m.scales[:3], 0.3) and not m.locations[:3].any()


def test_field_from_obj_triangles():
    obj = "v 0 0 0\nv 2 0 0\nv 2 0 -2\nv 0 0 -2\nvt 0.5 0.5\nf 1/1/1 2/1/1 3/1/1 4/1/1\n# comment\nf 1 2 3\n"
    verts, tris = gs.fields.parse_obj_triangles(obj)
    assert tris == [(0, 1, 2), (0, 2, 3), (0, 1, 2)]                       # quad fanned, then the triangle
    h = gs.fields.initFieldModel(obj)
    assert h.count == 3
    assert np.allclose(h.locations[:3], [4 / 3, 0, -2 / 3])
    assert np.allclose(h.scales[:3], [0.2 * 2.0, 0.2 * math.sqrt(8.0), 0.2 * 0.005])
    # normal of (v1-v0) x (v2-v0) = (2,0,0) x (2,0,-2) = (0,4,0) -> +Y; rotate +Z onto +Y: axis (-1,0,0)*1, angle 90 deg
    s = math.sin(math.pi / 4)
    assert np.allclose(h.rotations[:4], [-s, 0, 0, s], atol=1e-6)
    with pytest.raises(RuntimeError, match="Unexpected vertex count"):
        gs.fields.parse_obj_triangles("v 0 0 0\nf 1 1 1 1 1\n")


def test_gobj_round_trip(tmp_path):
    s = gs.synth.random_splats(50, 4, 3)
    h = gs.ModelSplatsHost.fromVectors(s["loc"], s["sh"], s["scale"], s["opac"], s["rot"])
    path = tmp_path / "m.gobj"
    gs.io.saveSplats(path, h)
    lines = open(path).read().splitlines()
    assert len(lines) == 5 * 50 and lines[0].startswith("v ") and lines[1].startswith("sh ") and lines[4].startswith("r ")
    assert lines[0] == "v " + " ".join("%g" % float(v) for v in s["loc"][:3])   # ostream default: 6 significant digits
    back = gs.io.loadSplats(path)
    assert (back.count, back.shCoeffs, back.shDegree, back.capacity) == (50, 4, 1, 1000000)
    for a, b, n in [(back.locations, s["loc"], 150), (back.shs, s["sh"], 600), (back.scales, s["scale"], 150),
                    (back.opacities, s["opac"], 50), (back.rotations, s["rot"], 200)]:
        assert np.allclose(a[:n], b, rtol=1e-5, atol=1e-7)                       # lossy by design
    bad = tmp_path / "bad.gobj"
    bad.write_text("v 0 0 0\nsh 1 2 3\ns 1 1 1\na 1\nr 1 0 0 0\nv 0 0 0\nsh 1 2 3 4 5 6\ns 1 1 1\na 1\nr 1 0 0 0\n")
    with pytest.raises(RuntimeError, match="Inconsistent SH degree"):
        gs.io.loadSplats(bad)


def test_settings_json_round_trip(tmp_path):
    f32 = lambda v: float(np.float32(v))
    p = gs.Project.initProject()
    assert p.sphere2.count == 0 and p.sphere2.fovDeg == 30.0 and p.sphere1.count == 16
    p.lrSh, p.iterations, p.sphere1.rotX, p.pathModel = 0.1, 3491, 123.0, "C:/models/a.obj"
    path = tmp_path / "settings.json"
    gs.io.saveSettings(path, p)
    text = path.read_text()
    j = json.loads(text)
    # `file << j` (src/ui/UiFrame.cpp:329): one compact line, keys in std::map order, floats as double(float member)
    assert "\n" not in text and ", " not in text and list(j) == sorted(j) and list(j["sphere1"]) == sorted(j["sphere1"])
    assert j["sphere1"] == {"count": 16, "distance": 10.0, "fovDeg": 60.0, "rotX": 123.0, "rotY": 0.0}
    assert "updateRule" not in j and j["intervalDensify"] == 200 and len(j) == 35 and j["lrSh"] == f32(0.1) != 0.1
    q = gs.io.loadSettings(path)
    assert q.lrSh == f32(0.1) and q.iterations == 3491 and q.sphere1.rotX == 123.0 and q.sphere2.count == 0 and q.pathModel == p.pathModel
    # from_json WITH_DEFAULT (src/Project.h:64): a key the file lacks gets a FRESH Project's value — also when loading into a used one
    path.write_text('{"lrScale": 0.5, "sphere2": {"count": 3}, "noSuchKey": [1, 2]}')
    used = gs.Project(lrSh=9.0, iterations=5)
    r = gs.io.loadSettings(path, used)
    assert r is used and r.lrScale == 0.5 and r.sphere2.count == 3 and r.sphere2.distance == 10.0 and r.lrSh == pytest.approx(0.0001) and r.iterations == 0
    for bad in ('{"iterations": "many"}', '{"previewTruth": 1}', '{"sphere1": 4}', '[1, 2]', '{"pathModel": 3}'):
        path.write_text(bad)
        with pytest.raises(RuntimeError):
            gs.io.loadSettings(path)


def test_two_sphere_camera_rig():
    p = gs.Project.initProject()
    p.sphere1.count, p.sphere2.count, p.sphere2.distance = 8, 4, 5.0
    cams = gs.camera.get_cameras_project(p)
    assert len(cams) == gs.camera.get_cameras_count(p) == 12
    assert np.allclose([np.linalg.norm(c.location) for c in cams[:8]], 10.0, atol=1e-4)
    assert np.allclose([np.linalg.norm(c.location) for c in cams[8:]], 5.0, atol=1e-4) and cams[8].fovDegY == 30.0
    base = np.array([c.location for c in cams[:8]])
    assert np.allclose(base, gs.camera.fibonacci_sphere(8, 10.0), atol=1e-5)  # rot 0 = identity
    p.sphere1.rotX = 90.0                                                      # rotX turns about +Y (src/Camera.cpp:40)
    rot = np.array([c.location for c in gs.camera.get_cameras_project(p)[:8]])
    assert np.allclose(rot[:, 1], base[:, 1], atol=1e-4)
    assert np.allclose(rot[:, 0], base[:, 2], atol=1e-4) and np.allclose(rot[:, 2], -base[:, 0], atol=1e-4)


def test_preview_camera():
    """Camera::getPreviewCamera, src/Camera.cpp:60-74: truth-index mode and the free (orbiting) camera."""
    p = gs.Project.initProject()
    assert (p.previewTruth, p.previewFreeOrbit, p.previewFreeDistance, p.previewFreeRotX) == (False, True, 10.0, 25.0)   # src/Project.h:50-58
    c = gs.camera.get_preview_camera(p)               # timer 0: no orbit; rotX = 25 degrees about +X lifts the camera
    assert np.allclose(c.location, [0.0, 10.0 * np.sin(np.radians(25.0)), -10.0 * np.cos(np.radians(25.0))], atol=1e-5)
    assert c.fovDegY == 60.0 and np.allclose(c.target, 0.0)
    p.previewTimer, p.previewFreeOrbitSpeed = 3.0, 0.5  # the orbit term is ADDED TO RADIANS as it is (src/Camera.cpp:68-69): 1.5 rad about +Y
    o = gs.camera.get_preview_camera(p)
    assert np.isclose(np.linalg.norm(o.location), 10.0, atol=1e-4) and np.isclose(o.location[1], c.location[1], atol=1e-5)
    ang = np.arctan2(o.location[0], o.location[2]) - np.arctan2(c.location[0], c.location[2])
    assert np.isclose((ang + np.pi) % (2 * np.pi) - np.pi, 1.5, atol=1e-4)
    p.previewFreeOrbit = False
    assert np.allclose(gs.camera.get_preview_camera(p).location, c.location, atol=1e-6)
    p.previewTruth, p.previewTruthIndex, p.sphere1.count = True, 2, 4
    t = gs.camera.get_preview_camera(p)
    assert np.array_equal(t.location, gs.camera.get_cameras_project(p)[2].location) and t.fovDegY == p.sphere1.fovDeg
    p.previewTruthIndex = 4                            # getCameras(project).at(index) throws past the end
    with pytest.raises(IndexError):
        gs.camera.get_preview_camera(p)


class _FakeTrainer:
    def __init__(self):
        self.calls = []

    def captureTruths(self, cameras, fw, fb):
        self.calls.append(("capture", len(cameras)))

    def train(self, project, densify):
        project.iterations += 1
        self.calls.append(("train", bool(densify)))


def test_auto_train_driver_loop():
    p = gs.Project.initProject()
    p.sphere1.count, p.intervalCapture, p.intervalDensify = 3, 4, 6
    tr = _FakeTrainer()
    now = [0.0]
    drv = gs.driver.AutoTrainer(tr, p, capture=lambda cams: ([None] * len(cams), [None] * len(cams)), clock=lambda: now[0])
    ran = 0
    for tick in range(400):                      # host idle events 12.5 ms apart: budget 100 steps/s -> one step per tick
        now[0] += 0.0125
        ran += drv.update()
    assert ran == 400 and p.iterations == 400
    trains = [c for c in tr.calls if c[0] == "train"]
    assert [i for i, c in enumerate(trains) if c[1]] == list(range(0, 400, 6))           # densify when iterations % 6 == 0 (incl. 0)
    assert sum(1 for c in tr.calls if c[0] == "capture") == 100 and tr.calls[0] == ("capture", 3)
    assert 0 <= p.sphere1.rotX < 360 and p.sphere1.rotX != 0.0                           # spheres re-rotated before each capture
    # throttle: events 1 ms apart only accumulate 0.1 budget each -> one step per 10 events
    tr2, p2 = _FakeTrainer(), gs.Project.initProject()
    drv2 = gs.driver.AutoTrainer(tr2, p2, capture=lambda cams: ([None] * len(cams), [None] * len(cams)), clock=lambda: now[0])
    ran = 0
    for tick in range(1000):
        now[0] += 0.001
        ran += drv2.update()
    assert 95 <= ran <= 100
    drv2.autoTraining = False
    now[0] += 1.0
    assert drv2.update() is False


def test_gobj_short_and_malformed_lines_follow_the_stream_reader(tmp_path):
    """The reference reads `v`, `s`, `a`, `r` lines with a fixed number of `iss >> x` extractions and pushes every one
    (src/ui/UiFrame.cpp:404-434): a short line, or one holding something a stream does not read as a number, still yields
    3 / 1 / 4 values — zero from the first failed extraction on."""
    path = tmp_path / "short.gobj"
    path.write_text("v 1 2\nsh 1 2 3\ns 1 nan 3\na\nr 1 0 0 0 9\n")
    m = gs.io.loadSplats(path)
    assert m.count == 1
    assert list(m.locations[:3]) == [1.0, 2.0, 0.0] and list(m.scales[:3]) == [1.0, 0.0, 0.0]
    assert m.opacities[0] == 0.0 and list(m.rotations[:4]) == [1.0, 0.0, 0.0, 0.0]


def test_cpp_extras_header_and_gobj_interop(tmp_path):
    """include/gsplat_extras.hpp (C++) against the Python mirror: a .gobj written by Python and copied by C++ equals
    Python's own copy byte for byte; the C++ grid and one-splat-per-triangle fields equal fields.py's."""
    import os
    import subprocess
    from gsplat_amd import capi
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "extras_cpu"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "cpp", "extras_cpu.cpp"),
                           "-o", str(exe), capi.LIB_PATH, "-Wl,-rpath," + os.path.dirname(capi.LIB_PATH), "-Wl,-rpath,/opt/rocm/lib"])
    s = gs.synth.random_splats(60, 4, 17)
    s["opac"][7] = 1e-5
    s["loc"][0] = 123456.789
    gs.io.saveSplats(tmp_path / "py.gobj", gs.ModelSplatsHost.fromVectors(s["loc"], s["sh"], s["scale"], s["opac"], s["rot"]))
    obj = "v 0 0 0\nv 2 0 0\nv 2 0 -2\nv 0 0 -2\nv 1 3 1\nvt 0.5 0.5\nf 1/1/1 2/1/1 3/1/1 4/1/1\n# comment\nf 1 2 5\nf 3//2 5//2 4//2\n"
    (tmp_path / "mesh.obj").write_text(obj)
    # settings.json, Python side of the interop: a project with values that exercise the number layout (tiny, huge, integral, negative zero)
    ps = gs.Project.initProject()
    ps.lrSh, ps.lrLocation, ps.paramScaleMax, ps.previewTimer, ps.previewFreeRotY, ps.iterations = 0.1, 1.0e-7, 123456.0, 1.0e22, -0.0, 3491
    ps.previewTruth, ps.previewFreeOrbit, ps.renderResX = True, False, 4096
    ps.pathModel, ps.perspective = 'C:\\models\\a "quoted" \u00e4.obj', "layout2|name=a;caption=\tb\n|"
    gs.io.saveSettings(tmp_path / "py_settings.json", ps)
    # 4000 float32 values over the whole finite range (random bit patterns, plus decades, integers and boundary cases of the layout rule)
    frng = np.random.default_rng(11)
    bits = frng.integers(0, 2 ** 32, 3000, dtype=np.uint64).astype(np.uint32)
    fl = bits.view(np.float32)
    fl = np.concatenate([fl[np.isfinite(fl)], np.float32(10.0) ** np.arange(-38, 39, dtype=np.float32), np.arange(-20, 20, dtype=np.float32),
                         np.array([0.0, -0.0, 1e-4, 9.999e-5, 1e16, 9.9999e15, 123456.0, 0.1, 1.0 / 3.0, 16777216.0, 3.4028235e38, 1.1754944e-38, 1e-45], np.float32)])
    fl.astype(np.float32).tofile(tmp_path / "floats.bin")
    out = subprocess.run([str(exe), str(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0 and "extras ok" in out.stdout, (out.returncode, out.stderr)
    # C++ load -> save of the Python file = the Python file; the same project built in C++ = the same bytes; Python reads both back
    assert (tmp_path / "cpp_settings.json").read_bytes() == (tmp_path / "py_settings.json").read_bytes()
    assert (tmp_path / "cpp_made.json").read_bytes() == (tmp_path / "py_settings.json").read_bytes()
    back = gs.io.loadSettings(tmp_path / "cpp_made.json")
    assert back.pathModel == ps.pathModel and back.perspective == ps.perspective and back.iterations == 3491 and back.previewTruth and not back.previewFreeOrbit
    assert back.lrSh == float(np.float32(0.1)) and back.previewTimer == float(np.float32(1e22)) and back.sphere2.fovDeg == 30.0 and back.renderResX == 4096
    # the C++ number writer = Python's float repr (what json.dumps writes) for every one of those values
    got = (tmp_path / "floats.txt").read_text().split("\n")[:-1]
    want = [json.dumps(float(x)) for x in fl.astype(np.float32)]
    assert len(got) == len(want) and got == want, [(a, b) for a, b in zip(got, want) if a != b][:5]
    # cameras: the C++ rig and preview camera against camera.py (same formulas in fp32; sin / cos of two libms)
    pr = gs.Project.initProject()
    pr.sphere1.count, pr.sphere1.rotX, pr.sphere1.rotY = 5, 40.0, -15.0
    pr.sphere2.count, pr.sphere2.distance, pr.sphere2.fovDeg, pr.sphere2.rotX = 3, 6.0, 30.0, 200.0
    pr.previewTimer, pr.previewFreeRotY, pr.previewFreeDistance, pr.previewFreeFovDeg = 2.5, 30.0, 7.0, 45.0
    want = [c for c in gs.camera.get_cameras_project(pr)] + [gs.camera.get_preview_camera(pr)]
    pr.previewFreeOrbit = False
    want.append(gs.camera.get_preview_camera(pr))
    pr.previewTruth, pr.previewTruthIndex = True, 6
    want.append(gs.camera.get_preview_camera(pr))
    got = np.loadtxt(tmp_path / "cpp_cameras.txt")
    assert got.shape == (11, 4)
    assert np.allclose(got[:, :3], [c.location for c in want], atol=2e-5) and np.array_equal(got[:, 3], [c.fovDegY for c in want])
    # copy of a Python-written file: the same bytes as Python's own load -> save
    gs.io.saveSplats(tmp_path / "py_copy.gobj", gs.io.loadSplats(tmp_path / "py.gobj"))
    assert (tmp_path / "cpp_copy.gobj").read_bytes() == (tmp_path / "py_copy.gobj").read_bytes()
    # grid: C++ indexes the lattice, the mirror accumulates floats like the reference: same file
    ref = gs.fields.initFieldGrid()
    ref.count = 40
    gs.io.saveSplats(tmp_path / "py_grid.gobj", ref)
    assert (tmp_path / "cpp_grid.gobj").read_bytes() == (tmp_path / "py_grid.gobj").read_bytes()
    # one splat per triangle: same values to fp32 rounding (sqrt / acos / sin of two libms)
    h, r = gs.io.loadSplats(tmp_path / "cpp_mesh.gobj"), gs.fields.initFieldModel(obj)
    assert h.count == r.count == 4
    for name, w in (("locations", 3), ("scales", 3), ("opacities", 1), ("rotations", 4)):
        assert np.allclose(getattr(h, name)[:w * 4], getattr(r, name)[:w * 4], rtol=2e-5, atol=2e-6), name


def test_lossless_checkpoint_round_trip(tmp_path):
    """io.saveCheckpoint / loadCheckpoint: model bit for bit (the .gobj text format keeps 6 significant digits), the Adam moments
    and step counter, the Project incl. the build-side update-rule fields; loaded with allow_pickle=False."""
    P, M = 70, 4
    s = gs.synth.random_splats(P, M, 19)
    host = gs.ModelSplatsHost.fromVectors(s["loc"], s["sh"], s["scale"], s["opac"], s["rot"])
    rng = np.random.default_rng(2)
    m1 = rng.standard_normal((11 + 3 * M) * 128).astype(np.float32)
    m2 = np.abs(rng.standard_normal((11 + 3 * M) * 128)).astype(np.float32)
    p = gs.Project.initProject()
    p.lrSh, p.iterations, p.sphere1.rotX, p.updateRule, p.adamEps = 0.25, 1234, 77.0, 1, 1e-12
    gs.io.saveCheckpoint(tmp_path / "run.npz", host, m1, m2, 1234, p)
    back, b1, b2, steps, q = gs.io.loadCheckpoint(tmp_path / "run.npz")
    assert (back.count, back.capacity, back.shCoeffs, back.shDegree, steps) == (P, host.capacity, M, host.shDegree, 1234)
    for name, w in (("locations", 3), ("shs", 3 * M), ("scales", 3), ("opacities", 1), ("rotations", 4)):
        assert np.array_equal(np.asarray(getattr(back, name)[:w * P]).view(np.uint32), np.asarray(getattr(host, name)[:w * P]).view(np.uint32)), name
    assert np.array_equal(b1.view(np.uint32), m1.view(np.uint32)) and np.array_equal(b2.view(np.uint32), m2.view(np.uint32))
    assert (q.lrSh, q.iterations, q.sphere1.rotX, q.sphere2.count, q.updateRule, q.adamEps) == (0.25, 1234, 77.0, 0, 1, 1e-12)
    gs.io.saveCheckpoint(tmp_path / "bare.npz", host)          # no optimizer state, no project
    back2, n1, n2, steps2, q2 = gs.io.loadCheckpoint(tmp_path / "bare.npz")
    assert n1 is None and n2 is None and steps2 == 0 and q2 is None and back2.count == P


def test_gobj_and_checkpoint_round_trips_on_random_models(tmp_path):
    """Property checks of the two file formats on random models (hypothesis; counts 0 ... 40, every SH size, values from denormals to
    1e30 and both zeros): `.gobj` — 6 significant digits, src/ui/UiFrame.cpp:333-358 — is idempotent after its first rounding (save,
    load, save gives the first file byte for byte, and the loaded values are within 5e-6 relative of the originals); the lossless
    checkpoint gives back every bit, moments and step counter included."""
    from hypothesis import given, settings, strategies as st

    big = float(np.float32(1e30))
    finite = st.floats(min_value=-big, max_value=big, allow_nan=False, allow_infinity=False, width=32)

    @settings(max_examples=60, deadline=None)
    @given(st.integers(0, 40), st.sampled_from([1, 4, 9, 16]), st.randoms(use_true_random=False), st.data())
    def prop(count, M, rnd, data):
        n_floats = count * (11 + 3 * M)
        vals = np.array(data.draw(st.lists(finite, min_size=n_floats, max_size=n_floats)), np.float32)
        if count:
            vals[rnd.randrange(n_floats)] = -0.0
            vals[rnd.randrange(n_floats)] = np.float32(1e-42)       # a denormal
        parts = np.split(vals, np.cumsum([3 * count, 3 * M * count, 3 * count, count]))
        m = gs.ModelSplatsHost(max(count, 1), (M - 1) // 3, M)
        m.count = count
        m.locations[:3 * count], m.shs[:3 * M * count], m.scales[:3 * count], m.opacities[:count], m.rotations[:4 * count] = parts
        a, b, c = tmp_path / "a.gobj", tmp_path / "b.gobj", tmp_path / "c.npz"
        gs.io.saveSplats(a, m)
        if count:
            back = gs.io.loadSplats(a)
            assert back.count == count and back.shCoeffs == M
            gs.io.saveSplats(b, back)
            assert a.read_bytes() == b.read_bytes()
            for x, y in ((m.locations, back.locations), (m.shs, back.shs), (m.scales, back.scales), (m.opacities, back.opacities), (m.rotations, back.rotations)):
                k = min(x.size, y.size)
                assert np.allclose(y[:k], x[:k], rtol=5e-6, atol=1e-37)
        else:
            assert a.read_bytes() == b""
        m1 = np.array(data.draw(st.lists(finite, min_size=8, max_size=8)), np.float32)
        gs.io.saveCheckpoint(c, m, m1, m1[::-1].copy(), adam_steps=rnd.randrange(0, 10 ** 6))
        host, r1, r2, steps, _ = gs.io.loadCheckpoint(c)
        assert host.count == count and host.shCoeffs == M and host.capacity == m.capacity
        for x, y in ((m.locations, host.locations), (m.shs, host.shs), (m.scales, host.scales), (m.opacities, host.opacities), (m.rotations, host.rotations)):
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
        assert np.array_equal(r1.view(np.uint32), m1.view(np.uint32)) and np.array_equal(r2.view(np.uint32), m1[::-1].view(np.uint32))
    prop()
