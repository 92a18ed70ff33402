"""The C++ drop-in shim (include/gsplat_shim.hpp: ModelSplatsHost / ModelSplatsDevice / Trainer with the
reference's names) driven by a small C++ program, compared bit-for-bit with the Python mirror."""
import os
import subprocess

import numpy as np
import pytest

import gsplat_amd as gs
from gsplat_amd import capi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_shim_matches_python_mirror(tmp_path, orc):
    exe = tmp_path / "shim_step"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "shim_step.cpp"),
                           "-o", str(exe), capi.LIB_PATH, "-Wl,-rpath," + os.path.dirname(capi.LIB_PATH), "-Wl,-rpath,/opt/rocm/lib"])
    P, M, W, H, Cn, steps = 900, 4, 96, 80, 2, 3
    s = gs.synth.random_splats(P, M, 321)
    cams = gs.camera.get_cameras(Cn)
    views = gs.camera.train_views(cams, W, H)
    rng = np.random.default_rng(0)
    fw = [rng.integers(0, 2 ** 32, W * H, dtype=np.uint32) for _ in range(Cn)]
    fb = [rng.integers(0, 2 ** 32, W * H, dtype=np.uint32) for _ in range(Cn)]
    with open(tmp_path / "in.bin", "wb") as f:
        np.array([P, M, W, H, Cn, steps], np.int32).tofile(f)
        for k in ("loc", "sh", "scale", "opac", "rot"):
            s[k].tofile(f)
        for c in cams:
            np.concatenate([c.location, c.target, [c.fovDegY]]).astype(np.float32).tofile(f)
        for a in fw + fb:
            a.tofile(f)
    subprocess.check_call([str(exe), str(tmp_path / "in.bin"), str(tmp_path / "out.bin")])
    raw = np.fromfile(tmp_path / "out.bin", np.uint8)
    count = int(raw[:4].view(np.int32)[0])
    assert count == P
    o = 4
    blocks = raw[o:o + 160 * 2 * Cn].view(np.float32).reshape(2 * Cn, 40); o += 160 * 2 * Cn
    # Camera::getView / getProjection of the shim against the Python mirror's (different libm / product order: ~1 ulp)
    assert np.allclose(blocks, views, rtol=2e-6, atol=2e-6)
    loc = raw[o:o + 12 * P].view(np.float32); o += 12 * P
    opac = raw[o:o + 4 * P].view(np.float32); o += 4 * P
    frame = raw[o:o + 4 * W * H].view(np.uint32)
    # same run through the Python mirror
    host = gs.ModelSplatsHost.fromVectors(s["loc"], s["sh"], s["scale"], s["opac"], s["rot"])
    tr = gs.Trainer(W, H)
    tr.model = gs.ModelSplatsDevice(host)
    tr.captureTruths(cams, fw, fb, view_blocks=blocks)   # the very pass parameters the C++ run used: results must agree bit for bit
    proj = gs.Project()
    for _ in range(steps):
        tr.train(proj)
    back = gs.ModelSplatsHost.fromDevice(tr.model)
    assert np.array_equal(back.locations[:3 * P].view(np.uint32), loc.view(np.uint32))
    assert np.array_equal(back.opacities[:P].view(np.uint32), opac.view(np.uint32))
    fb_py = tr.render(W, H, 1.0, cams[0])
    # Trainer::render incl. the reference's tan_fovx quirk.  The two camera implementations differ by an ulp, so the two frames may differ in
    # bytes whose float sits on a k / 256 boundary: BOTH are held to the oracle's float image of the same render (the Python mirror's
    # camera), every differing byte explained by that float lying within the pixel tolerance of the boundary (util.unexplained_bytes)
    import math
    from util import unexplained_bytes, view_parts
    blk = gs.camera.view_block(cams[0], W, H, white=False)
    blk[35] = np.float32(math.tan(math.radians(W * cams[0].fovDegY / H) * 0.5))
    vp = view_parts(blk)
    final = {k: getattr(back, n)[:m * P] for k, n, m in (("loc", "locations", 3), ("sh", "shs", 3 * M), ("scale", "scales", 3), ("opac", "opacities", 1), ("rot", "rotations", 4))}
    want_img, _ = orc.Rasterizer(np.float32).forward(back.shDegree, M, vp["bg"], W, H, final["loc"], final["sh"], final["opac"], final["scale"], 1.0, final["rot"],
                                                     vp["view"], vp["proj"], vp["campos"], vp["tanx"], vp["tany"])
    for name, fr in (("C++ shim", frame), ("Python mirror", fb_py)):
        n_off, n_unexplained = unexplained_bytes(fr, want_img, W, H)
        assert n_unexplained == 0 and n_off <= 1e-3 * 3 * W * H, (name, n_off, n_unexplained)
    lv = lambda a: ((a.reshape(-1)[:, None] >> np.arange(0, 32, 8)) & 0xFF).astype(int)
    assert np.abs(lv(frame) - lv(fb_py)).max() <= 1
    tr.close()
