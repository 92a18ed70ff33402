"""Shared helpers for the parity tests: scenes, and a wrapper that drives the C-ABI rasterizer seam
(gs_rasterize_forward / gs_rasterize_backward) the way src/Trainer.cu:334-412 drives the reference's."""
import ctypes as C
import math

import numpy as np

import gsplat_amd as gs
from gsplat_amd import capi

REC_DTYPE = np.dtype([("x", "f4"), ("y", "f4"), ("conA", "f4"), ("conB", "f4"), ("conC", "f4"), ("opacity", "f4"),
                      ("r", "f4"), ("g", "f4"), ("b", "f4"), ("hx", "f4"), ("hy", "f4"), ("depth", "f4"),
                      ("radius", "i4"), ("flags", "u4"), ("rect_min", "u4"), ("rect_max", "u4")])


def make_scene(P, M, seed, W, H, n_cams=1, fov=60.0, distance=10.0):
    s = gs.synth.random_splats(P, M, seed)
    cams = gs.camera.get_cameras(n_cams, distance, fov)
    views = gs.camera.train_views(cams, W, H)
    return s, cams, views


def view_parts(block):
    return dict(view=block[0:16].copy(), proj=block[16:32].copy(), campos=block[32:35].copy(), tanx=float(block[35]),
                tany=float(block[36]), bg=block[37:40].copy())


class SeamRaster:
    """Device-side rasterizer pair through the C-ABI seam, reference argument order."""

    def __init__(self):
        self.L = capi.lib()
        self.chunks = {}
        self._cbs = []

    def _alloc(self, name):
        def cb(nbytes, user):
            buf = capi.DeviceBuffer(nbytes)
            self.chunks[name] = buf
            return buf.ptr.value
        f = capi.ALLOC_FN(cb)
        self._cbs.append(f)
        return f

    def forward(self, s, D, M, vp, W, H, mod=1.0):
        self.P, self.D, self.M, self.W, self.H, self.mod = s["opac"].size, D, M, W, H, mod
        P = self.P
        dev = lambda a: capi.DeviceBuffer.from_numpy(np.ascontiguousarray(a, np.float32))
        self.d = dict(loc=dev(s["loc"]), sh=dev(s["sh"]), scale=dev(s["scale"]), opac=dev(s["opac"]), rot=dev(s["rot"]),
                      view=dev(vp["view"]), proj=dev(vp["proj"]), campos=dev(vp["campos"]), bg=dev(vp["bg"]))
        self.vp = vp
        self.out = capi.DeviceBuffer(3 * W * H * 4)
        R = C.c_int(-1)
        d = self.d
        capi.check(self.L.gs_rasterize_forward(
            self._alloc("geometry"), None, self._alloc("binning"), None, self._alloc("image"), None, P, D, M, d["bg"].ptr, W,
            H, d["loc"].ptr, d["sh"].ptr, None, d["opac"].ptr, d["scale"].ptr, C.c_float(mod), d["rot"].ptr, None,
            d["view"].ptr, d["proj"].ptr, d["campos"].ptr, C.c_float(vp["tanx"]), C.c_float(vp["tany"]), 0, self.out.ptr,
            None, 1, C.byref(R)))
        self.R = R.value
        return self.out.to_numpy(np.float32).reshape(3, H, W), self.R

    def field(self, chunk, name, dtype):
        off, nb = C.c_size_t(), C.c_size_t()
        capi.check(self.L.gs_raster_chunk_field(chunk.encode(), name.encode(), self.P, self.W, self.H, self.R,
                                                C.byref(off), C.byref(nb)))
        dtype = np.dtype(dtype)
        return self.chunks[chunk].to_numpy(dtype, nb.value // dtype.itemsize, off.value)

    def backward(self, dL_dpix, prefill=None):
        P, M = self.P, self.M
        sizes = dict(dL_dmean2D=3 * P, dL_dconic=4 * P, dL_dopacity=P, dL_dcolor=3 * P, dL_dmean3D=3 * P,
                     dL_dcov3D=6 * P, dL_dsh=3 * M * P, dL_dscale=3 * P, dL_drot=4 * P)
        g = {}
        for k, n in sizes.items():
            init = np.zeros(n, np.float32) if prefill is None else np.full(n, prefill.get(k, 0.0), np.float32)
            g[k] = capi.DeviceBuffer.from_numpy(init) if n else capi.DeviceBuffer(4)
        dpix = capi.DeviceBuffer.from_numpy(np.ascontiguousarray(dL_dpix, np.float32).reshape(-1))
        d, vp = self.d, self.vp
        capi.check(self.L.gs_rasterize_backward(
            P, self.D, M, self.R, d["bg"].ptr, self.W, self.H, d["loc"].ptr, d["sh"].ptr, None, d["scale"].ptr,
            C.c_float(self.mod), d["rot"].ptr, None, d["view"].ptr, d["proj"].ptr, d["campos"].ptr, C.c_float(vp["tanx"]),
            C.c_float(vp["tany"]), None, self.chunks["geometry"].ptr, self.chunks["binning"].ptr, self.chunks["image"].ptr,
            dpix.ptr, g["dL_dmean2D"].ptr, g["dL_dconic"].ptr, g["dL_dopacity"].ptr, g["dL_dcolor"].ptr,
            g["dL_dmean3D"].ptr, g["dL_dcov3D"].ptr, g["dL_dsh"].ptr, g["dL_dscale"].ptr, g["dL_drot"].ptr, 1))
        return {k: g[k].to_numpy(np.float32, sizes[k]) for k in sizes}


def oracle_forward(orc, s, D, M, vp, W, H, mod=1.0):
    r = orc.Rasterizer(np.float32)
    out, R = r.forward(D, M, vp["bg"], W, H, s["loc"], s["sh"], s["opac"], s["scale"], mod, s["rot"], vp["view"], vp["proj"],
                       vp["campos"], vp["tanx"], vp["tany"])
    return r, out, R


def assert_close_rel(name, got, want, rtol=1e-4, floor=None, max_bad_frac=0.0):
    """|got - want| <= rtol * max(|want|, floor); floor defaults to 1e-3 * max|want| (sums of many
    fp32 terms are only meaningful relative to the scale of the array)."""
    got = np.asarray(got, np.float64).reshape(-1)
    want = np.asarray(want, np.float64).reshape(-1)
    assert got.shape == want.shape, (name, got.shape, want.shape)
    if want.size == 0:
        return
    scale = np.abs(want).max()
    fl = (1e-3 * scale if floor is None else floor)
    tol = rtol * np.maximum(np.abs(want), fl) + 1e-30
    bad = np.abs(got - want) > tol
    frac = bad.mean()
    if frac > max_bad_frac:
        i = int(np.argmax(np.abs(got - want) / tol))
        raise AssertionError(f"{name}: {bad.sum()}/{bad.size} outside rtol={rtol} (worst idx {i}: got {got[i]!r} want {want[i]!r}, scale {scale!r})")
