#!/usr/bin/env python3
"""Bad arguments at every entry point of include/gsplat.h: each case runs in a process of its own (a crash is then a finding, not the
end of the run) and must come back with a non-zero gs_status and a message — never a signal.  A maintainer binding the library
behind the reference's classes gets std::runtime_error from the shim for each of these (the reference's own error type).
    gpurun -- 'python tools/abi_fuzz.py'            (all cases, one line each, exit code = number of crashes + wrongly accepted calls)
    python tools/abi_fuzz.py --case N               (one case in THIS process; what the driver spawns)
tests/test_gpu_trainer.py::test_c_abi_refuses_bad_arguments runs the same list in one process once it is clean."""
import argparse
import ctypes as C
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import gsplat_amd as gs  # noqa: E402
from gsplat_amd import capi  # noqa: E402

NULL = None
vp = C.c_void_p


def fixtures():
    """a live model, trainer (with and without views) and the usual scratch — every case gets fresh ones"""
    L = capi.lib()
    s = gs.synth.random_splats(50, 4, 3)
    host = gs.ModelSplatsHost.fromVectors(s["loc"], s["sh"], s["scale"], s["opac"], s["rot"])
    host.shDegree = 1
    W = H = 32
    tr = gs.Trainer(W, H)
    tr.model = gs.ModelSplatsDevice(host)
    bare = gs.Trainer(W, H)       # no model, no views
    cams = gs.camera.get_cameras(1)
    frame = np.zeros(W * H, np.uint32)
    tr.captureTruths(cams, [frame], [frame])
    tr._upload_views()
    return dict(L=L, tr=tr, t=tr.handle, bare=bare.handle, m=tr.model.handle, s=s, W=W, H=H, cams=cams, keep=[tr, bare, host, frame])


def cases():
    """(name, fn(fixtures) -> status).  Every call here is WRONG in some argument.  (Accepted by design and therefore not in the list: NULL
    hooks / NULL communicators — they uninstall the exchange; NULL optional outputs of gs_trainer_grad_buffer / _stage_times; a render
    without a model — the background; sh_coeffs = 5 with degree 1 — M is a stride, the reference's ModelSplatsHost takes any.)"""
    out = []
    add = lambda name: (lambda f: out.append((name, f)) or f)
    h = capi.gs_hyper()
    st = capi.gs_step_stats()
    fp = C.POINTER(C.c_float)
    buf16 = (C.c_float * 4096)()
    u32 = (C.c_uint32 * 4096)()
    outp = vp()
    n = C.c_size_t()
    i4 = [C.c_int() for _ in range(4)]

    add("device_malloc(NULL out)")(lambda x: x["L"].gs_device_malloc(NULL, 16))
    add("set_option(NULL name)")(lambda x: x["L"].gs_set_option(NULL, 1))
    add("set_option(unknown)")(lambda x: x["L"].gs_set_option(b"no_such_option", 1))
    add("hyper_defaults(NULL)")(lambda x: x["L"].gs_hyper_defaults(NULL))
    add("debug_wave_reduce9(NULL)")(lambda x: x["L"].gs_debug_wave_reduce9(NULL, NULL))
    add("debug_counters(NULL)")(lambda x: x["L"].gs_debug_counters(NULL, 0))
    arr = lambda x, k: x["s"][k].ctypes.data_as(fp)
    mc = lambda x, cap, D, M, cnt, loc=True, o=True: x["L"].gs_model_create(cap, D, M, cnt, arr(x, "loc") if loc else NULL, arr(x, "sh"), arr(x, "scale"), arr(x, "opac"),
                                                                         arr(x, "rot"), C.byref(outp) if o else NULL)
    add("model_create(capacity -1)")(lambda x: mc(x, -1, 1, 4, 0))
    add("model_create(count > capacity)")(lambda x: mc(x, 10, 1, 4, 20))
    add("model_create(count < 0)")(lambda x: mc(x, 10, 1, 4, -3))
    add("model_create(M = 0)")(lambda x: mc(x, 50, 0, 0, 50))
    add("model_create(NULL locations, count 50)")(lambda x: mc(x, 50, 1, 4, 50, loc=False))
    add("model_create(NULL out)")(lambda x: mc(x, 50, 1, 4, 50, o=False))
    add("model_clone(NULL src)")(lambda x: x["L"].gs_model_clone(NULL, C.byref(outp)))
    add("model_clone(NULL out)")(lambda x: x["L"].gs_model_clone(x["m"], NULL))
    add("model_download(NULL model)")(lambda x: x["L"].gs_model_download(NULL, buf16, buf16, buf16, buf16, buf16))
    add("model_info(NULL model)")(lambda x: x["L"].gs_model_info(NULL, C.byref(i4[0]), C.byref(i4[1]), C.byref(i4[2]), C.byref(i4[3])))
    add("trainer_create(0 x 0)")(lambda x: x["L"].gs_trainer_create(0, 0, C.byref(outp)))
    add("trainer_create(-5 x 10)")(lambda x: x["L"].gs_trainer_create(-5, 10, C.byref(outp)))
    add("trainer_create(NULL out)")(lambda x: x["L"].gs_trainer_create(16, 16, NULL))
    add("trainer_set_model(NULL trainer)")(lambda x: x["L"].gs_trainer_set_model(NULL, x["m"]))
    view = capi.gs_view()
    truth = (C.POINTER(C.c_uint32) * 2)()
    add("trainer_set_views(NULL trainer)")(lambda x: x["L"].gs_trainer_set_views(NULL, 1, C.byref(view), truth, 0, 1))
    add("trainer_set_views(n = -1)")(lambda x: x["L"].gs_trainer_set_views(x["t"], -1, C.byref(view), truth, 0, 1))
    add("trainer_set_views(NULL views, n = 2)")(lambda x: x["L"].gs_trainer_set_views(x["t"], 2, NULL, truth, 0, 2))
    add("trainer_set_views(NULL truth table)")(lambda x: x["L"].gs_trainer_set_views(x["t"], 1, C.byref(view), NULL, 0, 1))
    add("trainer_set_views(NULL truth image)")(lambda x: x["L"].gs_trainer_set_views(x["t"], 1, C.byref(view), truth, 0, 1))
    add("trainer_set_views(total_samples < n_views)")(lambda x: x["L"].gs_trainer_set_views(x["t"], 2, (capi.gs_view * 2)(), truth, 0, 1))
    add("trainer_step(NULL trainer)")(lambda x: x["L"].gs_trainer_step(NULL, C.byref(h), 0, NULL))
    add("trainer_step(NULL hyper)")(lambda x: x["L"].gs_trainer_step(x["t"], NULL, 0, NULL))
    add("trainer_step(no model, no views)")(lambda x: x["L"].gs_trainer_step(x["bare"], C.byref(h), 0, C.byref(st)))
    add("trainer_accumulate(NULL)")(lambda x: x["L"].gs_trainer_accumulate(NULL, NULL))
    add("trainer_accumulate(no views)")(lambda x: x["L"].gs_trainer_accumulate(x["bare"], NULL))
    add("trainer_grad_buffer(NULL trainer)")(lambda x: x["L"].gs_trainer_grad_buffer(NULL, C.byref(outp), C.byref(n)))
    add("trainer_apply(NULL trainer)")(lambda x: x["L"].gs_trainer_apply(NULL, C.byref(h), 0, NULL))
    add("trainer_apply(NULL hyper)")(lambda x: x["L"].gs_trainer_apply(x["t"], NULL, 0, NULL))
    add("trainer_apply(no model)")(lambda x: x["L"].gs_trainer_apply(x["bare"], C.byref(h), 0, NULL))
    add("trainer_set_option(NULL trainer)")(lambda x: x["L"].gs_trainer_set_option(NULL, b"cull", 1))
    add("trainer_set_option(NULL name)")(lambda x: x["L"].gs_trainer_set_option(x["t"], NULL, 1))
    add("trainer_set_option(unknown)")(lambda x: x["L"].gs_trainer_set_option(x["t"], b"no_such_option", 1))
    add("trainer_adam_state(NULL trainer)")(lambda x: x["L"].gs_trainer_adam_state(NULL, C.byref(outp), C.byref(outp), C.byref(n), C.byref(i4[0])))
    add("trainer_set_adam_state(NULL trainer)")(lambda x: x["L"].gs_trainer_set_adam_state(NULL, buf16, buf16, 10, 1, 0))
    add("trainer_set_adam_state(NULL moments)")(lambda x: x["L"].gs_trainer_set_adam_state(x["t"], NULL, NULL, 10, 1, 0))
    add("trainer_set_adam_state(wrong size)")(lambda x: x["L"].gs_trainer_set_adam_state(x["t"], buf16, buf16, 7, 1, 0))
    add("trainer_set_adam_state(steps -1)")(lambda x: x["L"].gs_trainer_set_adam_state(x["t"], buf16, buf16, 10, -1, 0))
    add("trainer_set_allreduce(NULL trainer)")(lambda x: x["L"].gs_trainer_set_allreduce(NULL, NULL, NULL))
    cb = capi.ALLREDUCE_FN(lambda *a: 0)
    add("trainer_set_sharded_update(NULL trainer)")(lambda x: x["L"].gs_trainer_set_sharded_update(NULL, cb, cb, NULL, 0, 2))
    add("trainer_set_sharded_update(rank >= world)")(lambda x: x["L"].gs_trainer_set_sharded_update(x["t"], cb, cb, NULL, 3, 2))
    add("trainer_set_sharded_update(world 0)")(lambda x: x["L"].gs_trainer_set_sharded_update(x["t"], cb, cb, NULL, 0, 0))
    add("trainer_set_compact_exchange(NULL trainer)")(lambda x: x["L"].gs_trainer_set_compact_exchange(NULL, cb, cb, NULL, 0, 2, 2, buf16))
    add("trainer_set_compact_exchange(rank >= world)")(lambda x: x["L"].gs_trainer_set_compact_exchange(x["t"], cb, cb, NULL, 2, 2, 2, buf16))
    add("trainer_set_compact_exchange(0 cameras)")(lambda x: x["L"].gs_trainer_set_compact_exchange(x["t"], cb, cb, NULL, 0, 2, 0, buf16))
    add("trainer_set_compact_exchange(fewer cameras than ranks)")(lambda x: x["L"].gs_trainer_set_compact_exchange(x["t"], cb, cb, NULL, 0, 2, 1, NULL))
    add("debug_hbm_copy_rate(NULL out)")(lambda x: x["L"].gs_debug_hbm_copy_rate(1 << 20, 1, None))
    add("debug_hbm_copy_rate(bytes not a multiple of 16)")(lambda x: x["L"].gs_debug_hbm_copy_rate(1000, 1, C.byref(C.c_double())))
    add("debug_hbm_copy_rate(0 repeats)")(lambda x: x["L"].gs_debug_hbm_copy_rate(1 << 20, 0, C.byref(C.c_double())))
    add("trainer_get_stream(NULL trainer)")(lambda x: x["L"].gs_trainer_get_stream(NULL, C.byref(outp)))
    add("trainer_get_stream(NULL out)")(lambda x: x["L"].gs_trainer_get_stream(x["t"], NULL))
    add("trainer_synchronize(NULL)")(lambda x: x["L"].gs_trainer_synchronize(NULL))
    add("trainer_set_profiling(NULL)")(lambda x: x["L"].gs_trainer_set_profiling(NULL, 1))
    add("trainer_stage_times(NULL trainer)")(lambda x: x["L"].gs_trainer_stage_times(NULL, NULL, NULL))
    add("trainer_read_image(NULL trainer)")(lambda x: x["L"].gs_trainer_read_image(NULL, 0, buf16))
    add("trainer_read_image(view -1)")(lambda x: x["L"].gs_trainer_read_image(x["t"], -1, buf16))
    add("trainer_read_image(view 99)")(lambda x: x["L"].gs_trainer_read_image(x["t"], 99, buf16))
    add("trainer_read_image(NULL out)")(lambda x: x["L"].gs_trainer_read_image(x["t"], 0, NULL))
    rend = lambda x, t, fb, w, hh, v: x["L"].gs_trainer_render(t, fb, 0, w, hh, C.c_float(1.0), v)
    add("trainer_render(NULL trainer)")(lambda x: rend(x, NULL, u32, 16, 16, C.byref(view)))
    add("trainer_render(NULL framebuffer)")(lambda x: rend(x, x["t"], NULL, 16, 16, C.byref(view)))
    add("trainer_render(0 x 0)")(lambda x: rend(x, x["t"], u32, 0, 0, C.byref(view)))
    add("trainer_render(NULL view)")(lambda x: rend(x, x["t"], u32, 16, 16, NULL))
    add("image_float_to_int(NULL)")(lambda x: x["L"].gs_image_float_to_int(NULL, NULL, 4, 4))
    add("image_int_to_loss(NULL)")(lambda x: x["L"].gs_image_int_to_loss(NULL, NULL, NULL, 4, 4))
    off = C.c_size_t()
    add("raster_chunk_field(NULL names)")(lambda x: x["L"].gs_raster_chunk_field(NULL, NULL, 10, 16, 16, 10, C.byref(off), C.byref(n)))
    add("raster_chunk_field(unknown field)")(lambda x: x["L"].gs_raster_chunk_field(b"geometry", b"nope", 10, 16, 16, 10, C.byref(off), C.byref(n)))
    add("raster_chunk_field(NULL outs)")(lambda x: x["L"].gs_raster_chunk_field(b"geometry", b"record", 10, 16, 16, 10, NULL, NULL))
    R = C.c_int()
    keep = []

    def dev(x):      # the seam takes DEVICE pointers: a zero-filled device buffer stands in for every array the call must not reach
        if "dev" not in x:
            x["dev"] = capi.DeviceBuffer(1 << 20)
            capi.check(x["L"].gs_memset_d(x["dev"].ptr, 0, 1 << 20))
        return x["dev"].ptr
    fwd = lambda x, alloc, P, W, means=True: x["L"].gs_rasterize_forward(
        alloc, NULL, alloc, NULL, alloc, NULL, P, 1, 4, dev(x), W, 16, dev(x) if means else NULL, dev(x), NULL, dev(x), dev(x), C.c_float(1.0), dev(x),
        NULL, dev(x), dev(x), dev(x), C.c_float(1.0), C.c_float(1.0), 0, dev(x), NULL, 1, C.byref(R))

    def alloc_ok(x):
        def cb2(nbytes, user):
            b = capi.DeviceBuffer(max(nbytes, 4)); keep.append(b); return b.ptr.value
        f = capi.ALLOC_FN(cb2); keep.append(f); return f
    add("rasterize_forward(NULL allocators)")(lambda x: fwd(x, C.cast(NULL, capi.ALLOC_FN), 10, 16))
    add("rasterize_forward(P = -1)")(lambda x: fwd(x, alloc_ok(x), -1, 16))
    add("rasterize_forward(width 0)")(lambda x: fwd(x, alloc_ok(x), 10, 0))
    add("rasterize_forward(NULL means, P = 10)")(lambda x: fwd(x, alloc_ok(x), 10, 16, means=False))
    add("rasterize_forward(allocator returns NULL)")(lambda x: fwd(x, keep.append(capi.ALLOC_FN(lambda nb, u: None)) or keep[-1], 10, 16))
    add("rasterize_backward(NULL chunks)")(lambda x: x["L"].gs_rasterize_backward(
        10, 1, 4, 5, dev(x), 16, 16, dev(x), dev(x), NULL, dev(x), C.c_float(1.0), dev(x), NULL, dev(x), dev(x), dev(x), C.c_float(1.0), C.c_float(1.0), NULL, NULL, NULL, NULL,
        dev(x), dev(x), dev(x), dev(x), dev(x), dev(x), dev(x), dev(x), dev(x), dev(x), 1))
    add("comm_unique_id(NULL)")(lambda x: x["L"].gs_comm_unique_id(NULL))
    cid = (C.c_char * capi.GS_COMM_ID_BYTES)()
    add("comm_create(NULL id)")(lambda x: x["L"].gs_comm_create(NULL, 0, 1, C.byref(outp)))
    add("comm_create(rank >= ranks)")(lambda x: x["L"].gs_comm_create(cid, 5, 2, C.byref(outp)))
    add("comm_create(NULL out)")(lambda x: x["L"].gs_comm_create(cid, 0, 1, NULL))
    add("trainer_attach_comm(NULL, NULL)")(lambda x: x["L"].gs_trainer_attach_comm(NULL, NULL))
    return out


def run_case(i, fx=None):
    name, fn = cases()[i]
    fx = fx or fixtures()
    status = fn(fx)
    msg = (fx["L"].gs_last_error() or b"").decode(errors="replace")
    return name, status, msg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", type=int, default=-1)
    a = ap.parse_args()
    if a.case >= 0:
        name, status, msg = run_case(a.case)
        print(f"{status}\t{msg}")
        return 0
    bad = 0
    for i, (name, _) in enumerate(cases()):
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--case", str(i)], capture_output=True, text=True, timeout=120)
        if p.returncode != 0:
            bad += 1
            print(f"CRASH   {name}: exit {p.returncode} {p.stderr.strip().splitlines()[-1][:200] if p.stderr.strip() else ''}")
            continue
        status, _, msg = p.stdout.strip().splitlines()[-1].partition("\t")
        if int(status) == 0:
            bad += 1
            print(f"ACCEPTED {name}")
        else:
            print(f"ok      {name}: {status} {msg[:110]}")
    print(f"{bad} findings")
    return min(bad, 100)


if __name__ == "__main__":
    sys.exit(main())
