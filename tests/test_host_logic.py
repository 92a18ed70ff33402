"""CPU tests of the host-side mirror: ModelSplatsHost semantics (src/ModelSplatsHost.cpp), Project
-> gs_hyper mapping (src/Project.h:26-41), PCG32 inputs, and that the C-ABI library loads and
exports every symbol include/gsplat.h declares (no compute calls here: this box has no GPU)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import gsplat_amd as gs
from gsplat_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "gsplat.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(gs_[a-z0-9_]+)\s*\(", header))
    declared -= {"gs_alloc_fn", "gs_allreduce_fn"}
    assert len(declared) >= 35
    L = capi.lib()
    missing = [n for n in sorted(declared) if not hasattr(L, n)]
    assert not missing, missing
    assert set(capi.SYMBOLS) == declared


def test_struct_layouts_match_header():
    assert C.sizeof(capi.gs_view) == 160
    assert C.sizeof(capi.gs_hyper) == 18 * 4
    h = capi.hyper_defaults()   # src/Project.h:26-41 defaults
    p = gs.Project()
    got = p.hyper()
    for name, _ in capi.gs_hyper._fields_:
        assert getattr(h, name) == getattr(got, name), name
    assert abs(h.lr_location - 0.00005) < 1e-10 and abs(h.clone_distance - 1.6) < 1e-6 and h.update_rule == 0


def test_no_gpu_means_loud_failure_not_fallback():
    L = capi.lib()
    if L.gs_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(capi.GsError) as e:
        gs.Trainer(64, 64)
    assert e.value.status == -9 and "no CPU path" in str(e.value)
    with pytest.raises(capi.GsError):
        capi.DeviceBuffer(16)


def test_resolution_limits_are_argument_errors_not_crashes():
    """The binning keeps one LDS counter per 64x64-px super-tile (16384 of them = 8192 x 8192 px, the largest render the
    reference's UI offers, src/ui/tools/UiPanelToolsView.cpp:120,125): larger images are refused up front, with or
    without a GPU (the check precedes the device check).  8192 x 8192 itself is accepted (tests/test_gpu_fullsize.py
    renders it on the GPU)."""
    for w, h in [(0, 64), (64, -1), (16384, 16384), (8192, 8256), (16448, 4096)]:
        with pytest.raises(capi.GsError) as e:
            gs.Trainer(w, h)
        assert e.value.status == -1, (w, h, str(e.value))   # GS_ERR_INVALID_ARGUMENT
    assert "super-tiles" in str(e.value)


def test_model_host_constructor_validation():
    P, M = 5, 4
    s = gs.synth.random_splats(P, M, 1)
    h = gs.ModelSplatsHost.fromVectors(s["loc"], s["sh"], s["scale"], s["opac"], s["rot"])
    assert (h.capacity, h.count, h.shCoeffs, h.shDegree) == (1000000, P, 4, 1)
    h16 = gs.ModelSplatsHost.fromVectors(s["loc"], np.zeros(3 * 16 * P), s["scale"], s["opac"], s["rot"])
    assert h16.shDegree == 5   # the reference's (M-1)/3 quirk, src/ModelSplatsHost.cpp:36
    with pytest.raises(RuntimeError, match="Inconsistent feature dimensions"):
        gs.ModelSplatsHost.fromVectors(s["loc"], s["sh"], s["scale"][:-3], s["opac"], s["rot"])
    with pytest.raises(RuntimeError, match="Inconsistent feature dimensions"):
        gs.ModelSplatsHost.fromVectors(s["loc"], s["sh"][:-1], s["scale"], s["opac"], s["rot"])


def test_model_host_pushback_and_copy():
    h = gs.ModelSplatsHost(2, 1, 4)
    h.pushBack([1, 2, 3], list(range(12)), [0.1, 0.2, 0.3], 0.5, [0, 0, 0, 1])
    h.pushBack([4, 5, 6], [0.0] * 12, [0.05] * 3, 1.0, [0, 0, 0, 1])
    with pytest.raises(RuntimeError, match="ran out of capacity"):
        h.pushBack([0, 0, 0], [0.0] * 12, [0.05] * 3, 1.0, [0, 0, 0, 1])
    h.copy(1, 0)
    assert np.array_equal(h.locations[3:6], [1, 2, 3]) and np.array_equal(h.shs[12:24], np.arange(12))
    for bad in [(2, 0), (0, 2), (-1, 0)]:
        with pytest.raises(RuntimeError, match="incorrect bounds"):
            h.copy(*bad)
    with pytest.raises(IndexError):
        gs.ModelSplatsHost(1, 1, 4).pushBack([0, 0, 0], [0.0] * 11, [0.05] * 3, 1.0, [0, 0, 0, 1])


def test_pcg32_reference_vector_and_ranges():
    # first outputs of the PCG reference implementation's demo: pcg32_srandom(42, 54)
    assert [hex(x) for x in gs.synth.pcg32(42, 54, 6)] == ['0xa15c02b7', '0x7b47f409', '0xba1d3330', '0x83d2f293', '0xbfa4784b', '0xcbed606e']
    s = gs.synth.random_splats(5000, 16, gs.synth.seed_for(3))
    assert s["loc"].min() >= -4 and s["loc"].max() <= 4 and s["scale"].min() >= 0.01 and s["scale"].max() <= 0.06
    assert s["opac"].min() >= 0.1 and s["opac"].max() <= 1.0
    q = s["rot"].reshape(-1, 4)
    assert np.allclose(np.linalg.norm(q, axis=1), 1.0, atol=1e-6)
    sh = s["sh"].reshape(-1, 16, 3)
    assert np.abs(sh[:, 1:]).max() <= 0.2 and np.abs(sh[:, 0]).max() <= 1.5 and np.abs(sh[:, 0]).max() > 1.0
    a = gs.synth.random_splats(100, 4, 7)
    b = gs.synth.random_splats(100, 4, 7)
    assert all(np.array_equal(a[k], b[k]) for k in ("loc", "sh", "scale", "opac", "rot"))


def test_source_stamp_ignores_comments_only(tmp_path):
    """tools/source_stamp.py identifies the kernel sources a counter collection belongs to: a comment or white-space edit
    keeps the stamp, a code edit changes it (bench.py reports counters only for matching stamps)."""
    import shutil
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from source_stamp import source_stamp
    root = tmp_path / "r"
    shutil.copytree(os.path.join(ROOT, "gaussian-splatterer_amd", "csrc"), root / "gaussian-splatterer_amd" / "csrc", ignore=shutil.ignore_patterns("_obj", "*.o"))
    os.makedirs(root / "include")
    shutil.copy(os.path.join(ROOT, "include", "gsplat.h"), root / "include" / "gsplat.h")
    base = source_stamp(str(root))
    assert base == source_stamp(ROOT)
    f = root / "gaussian-splatterer_amd" / "csrc" / "k_update.hip"
    text = f.read_text()
    f.write_text("// a new comment\n" + text.replace("\n", "\n   ", 3) + "\n/* trailing\n block */\n")
    assert source_stamp(str(root)) == base
    f.write_text(text)
    f = root / "gaussian-splatterer_amd" / "csrc" / "gs_internal.h"     # the update rule both update kernels share
    text = f.read_text()
    assert "1.0f - u.b1" in text
    f.write_text(text.replace("1.0f - u.b1", "1.0f - u.b2", 1))
    assert source_stamp(str(root)) != base


def test_cpp_shim_and_its_reference_call_sites_compile():
    """include/gsplat_shim.hpp + the program that uses it exactly like the reference's UI does (tests/cpp/shim_step.cpp:
    `delete trainer->model; trainer->model = new ...`, `train(project, densify)`, `render(fb, w, h, scale, camera)`,
    `truthCameras`) compile with a plain host compiler (the run itself needs the GPU: tests/test_gpu_shim.py)."""
    import subprocess
    for src in ("shim_step.cpp", "extras_cpu.cpp", "shim_hyper_overload.cpp"):
        subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"),
                               os.path.join(ROOT, "tests", "cpp", src)])
