"""ctypes wrapper around oracle/_build/libgs_oracle.so.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
shipped HIP path (gaussian-splatterer_amd/) never does.  See gs_oracle.cpp's header: the oracle is
a CPU restatement of the reference step (src/Trainer.cu) and parity is UNPINNED by the reference
(it has no tests or fixtures); the pins are fp64 finite differences, closed-form cases and
invariants in tests/test_oracle_*.py.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libgs_oracle.so")
_lib = None

f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)
u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int)


def build(force=False):
    """Compile the oracle with g++ (oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(
            os.path.join(_HERE, "gs_oracle.cpp")):
        subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def usable_cpus():
    """CPUs this process may use: the affinity mask, cut to the cgroup's CPU quota (cpu.max of cgroup v2, cfs_quota of v1)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, -(-quota // period)))
        except (OSError, ValueError):
            pass
    return n


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        if "OMP_NUM_THREADS" not in os.environ:      # the user's word stands; otherwise one thread per CPU the process may use
            _lib.orc_set_num_threads(C.c_int(usable_cpus()))
        _lib.orc_state_new.restype = C.c_void_p
        _lib.orc_state_free.argtypes = [C.c_void_p]
        _lib.orc_get.restype = C.c_long
        _lib.orc_get.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_long]
        _lib.orc_num_rendered.argtypes = [C.c_void_p]
        _lib.orc_densify.restype = C.c_int
    return _lib


def _p(a, ty):
    return a.ctypes.data_as(ty) if a is not None else None


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


_FIELDS = {
    "depth": np.float32, "means2D": np.float32, "cov3D": np.float32, "conic_opacity": np.float32,
    "rgb": np.float32, "radii": np.int32, "tiles_touched": np.uint32, "point_offsets": np.uint32,
    "rect": np.uint32, "clamped": np.uint8, "keys_unsorted": np.uint64, "keys": np.uint64,
    "vals_unsorted": np.uint32, "point_list": np.uint32, "ranges": np.uint32, "final_T": np.float32,
    "n_contrib": np.uint32, "margin": np.float32, "splat_margin": np.float32,
}


class Rasterizer:
    """The CudaRasterizer::Rasterizer::forward/backward pair as the reference calls it
    (src/Trainer.cu:334-360, :378-412), on the CPU.  dtype float32 = the oracle; float64 = FD checks."""

    def __init__(self, dtype=np.float32):
        self.dt = np.dtype(dtype)
        self.h = C.c_void_p(lib().orc_state_new())
        self.P = 0

    def __del__(self):
        try:
            if self.h:
                lib().orc_state_free(self.h)
                self.h = None
        except Exception:
            pass

    def forward(self, D, M, bg, W, H, means, shs, opac, scales, mod, rots, view, proj, campos, tanx, tany):
        dt = self.dt
        self.args = dict(D=D, M=M, bg=_c(bg, dt), W=W, H=H, means=_c(means, dt).reshape(-1),
                         shs=_c(shs, dt).reshape(-1), opac=_c(opac, dt).reshape(-1),
                         scales=_c(scales, dt).reshape(-1), mod=mod, rots=_c(rots, dt).reshape(-1),
                         view=_c(view, dt).reshape(-1), proj=_c(proj, dt).reshape(-1), campos=_c(campos, dt),
                         tanx=tanx, tany=tany)
        a = self.args
        P = a["opac"].size
        self.P = P
        out = np.zeros((3, H, W), dtype=dt)
        if dt == np.float32:
            fn, pt, sc = lib().orc_forward_f32, f32p, C.c_float
        else:
            fn, pt, sc = lib().orc_forward_f64, f64p, C.c_double
        R = fn(self.h, C.c_int(P), C.c_int(D), C.c_int(M), _p(a["bg"], pt), C.c_int(W), C.c_int(H),
               _p(a["means"], pt), _p(a["shs"], pt), _p(a["opac"], pt), _p(a["scales"], pt), sc(mod),
               _p(a["rots"], pt), _p(a["view"], pt), _p(a["proj"], pt), _p(a["campos"], pt), sc(tanx), sc(tany),
               _p(out, pt))
        self.R = R
        return out, R

    def get(self, name):
        assert self.dt == np.float32
        dt = _FIELDS[name]
        cap = max(self.P, 1) * 64 + (self.args["W"] * self.args["H"] + 1) * 16 + (self.R + 1) * 16
        buf = np.zeros(cap, dtype=np.uint8)
        n = lib().orc_get(self.h, name.encode(), buf.ctypes.data_as(C.c_void_p), C.c_long(cap))
        if n < 0:
            raise KeyError(name)
        return buf[:n].view(dt).copy()

    def backward(self, dL_dpix, want_abs=False, flip_margin=None, want_cond=False, power_ulps=0.0):
        """flip_margin (fp32 only): also return g["flip9"][P, 9], the admissible deviation of an implementation that
        takes the other branch at every blend decision within that relative margin of its threshold (gs_oracle.cpp,
        render_backward).  want_cond (with flip_margin): also g["cond9"][P, 9], the conditioning of every sum in units of
        2^-24 (gs_oracle.cpp, pixel_cond): what the cancellations inside the exponent, dG/dmean and dL/dalpha cost ANY fp32
        evaluation order.  power_ulps > 0 (with want_cond): flip9 also covers the `power > 0: skip` decision of every pair whose power lies
        within that many units of its own conditioning (2^-24 m) of zero."""
        a = self.args
        dt = self.dt
        P, M = self.P, a["M"]
        g = dict(dL_dmean2D=np.zeros(3 * P, dt), dL_dconic=np.zeros(4 * P, dt), dL_dopacity=np.zeros(P, dt),
                 dL_dcolor=np.zeros(3 * P, dt), dL_dmean3D=np.zeros(3 * P, dt), dL_dcov3D=np.zeros(6 * P, dt),
                 dL_dsh=np.zeros(3 * M * P, dt), dL_dscale=np.zeros(3 * P, dt), dL_drot=np.zeros(4 * P, dt))
        dpix = _c(dL_dpix, dt).reshape(-1)
        if dt == np.float32:
            abs9 = np.zeros(9 * P, np.float64) if (want_abs or flip_margin is not None) else None
            flip9 = np.zeros(9 * P, np.float64) if flip_margin is not None else None
            fn = lib().orc_backward_f32 if flip_margin is None else lib().orc_backward_f32_flip
            extra = () if flip_margin is None else (_p(flip9, f64p), C.c_float(flip_margin))
            cond9 = None
            if want_cond:
                assert flip_margin is not None
                cond9 = np.zeros(9 * P, np.float64)
                fn, extra = lib().orc_backward_f32_cond, extra + (_p(cond9, f64p), C.c_float(power_ulps))
            fn(self.h, C.c_int(a["D"]), C.c_int(M), _p(a["bg"], f32p), _p(a["means"], f32p),
                                   _p(a["shs"], f32p), _p(a["scales"], f32p), C.c_float(a["mod"]),
                                   _p(a["rots"], f32p), _p(a["view"], f32p), _p(a["proj"], f32p),
                                   _p(a["campos"], f32p), C.c_float(a["tanx"]), C.c_float(a["tany"]),
                                   _p(dpix, f32p), _p(g["dL_dmean2D"], f32p), _p(g["dL_dconic"], f32p),
                                   _p(g["dL_dopacity"], f32p), _p(g["dL_dcolor"], f32p), _p(g["dL_dmean3D"], f32p),
                                   _p(g["dL_dcov3D"], f32p), _p(g["dL_dsh"], f32p), _p(g["dL_dscale"], f32p),
                                   _p(g["dL_drot"], f32p), _p(abs9, f64p), *extra)
            if cond9 is not None:
                g["cond9"] = cond9.reshape(P, 9)
            if abs9 is not None:
                g["abs9"] = abs9.reshape(P, 9)
            if flip9 is not None:
                g["flip9"] = flip9.reshape(P, 9)
        else:
            lib().orc_backward_f64(self.h, C.c_int(a["D"]), C.c_int(M), _p(a["bg"], f64p), _p(a["means"], f64p),
                                   _p(a["shs"], f64p), _p(a["scales"], f64p), C.c_double(a["mod"]),
                                   _p(a["rots"], f64p), _p(a["view"], f64p), _p(a["proj"], f64p),
                                   _p(a["campos"], f64p), C.c_double(a["tanx"]), C.c_double(a["tany"]),
                                   _p(dpix, f64p), _p(g["dL_dmean2D"], f64p), _p(g["dL_dconic"], f64p),
                                   _p(g["dL_dopacity"], f64p), _p(g["dL_dcolor"], f64p), _p(g["dL_dmean3D"], f64p),
                                   _p(g["dL_dcov3D"], f64p), _p(g["dL_dsh"], f64p), _p(g["dL_dscale"], f64p),
                                   _p(g["dL_drot"], f64p))
        return g


    # --- the reference's own fp32 atomicAdd summation (gs_oracle.cpp, atomic_prepare / atomic_sums) ---
    def atomic_prepare(self, dL_dpix):
        """Keep every term upstream's backward render kernel would atomicAdd for this dL_dpix (state of the last forward).
        Returns their number."""
        assert self.dt == np.float32
        a = self.args
        dpix = _c(dL_dpix, np.float32).reshape(-1)
        assert dpix.size == 3 * a["W"] * a["H"]
        L = lib()
        L.orc_atomic_prepare.restype = C.c_uint64
        return int(L.orc_atomic_prepare(self.h, _p(a["bg"], f32p), _p(dpix, f32p)))

    def atomic_sums(self, seed, mode=0):
        """The nine pixel-stage sums [P, 9] (layout of chain()) added in fp32 in the order (seed, mode) names: one admissible
        run of the reference.  mode 0: one random order over all of a splat's (tile, pixel) terms; 1: random tile order, random
        pixel order inside each tile; 2: ascending tile / raster pixel order; 3: its reverse."""
        sums9 = np.zeros((self.P, 9), np.float32)
        lib().orc_atomic_sums(self.h, C.c_uint64(seed), C.c_int(mode), _p(sums9, f32p))
        return sums9

    def atomic_release(self):
        lib().orc_atomic_release(self.h)

    def atomic_backward(self, seed, mode=0):
        """One admissible run of the reference's backward (after atomic_prepare): the sums in the (seed, mode) order, then the
        unchanged per-splat chain.  Same keys as backward()."""
        P = self.P
        sums9 = self.atomic_sums(seed, mode)
        g = chain(self, sums9)
        g["dL_dcolor"] = np.ascontiguousarray(sums9[:, 0:3]).reshape(-1)
        m2 = np.zeros((P, 3), np.float32); m2[:, :2] = sums9[:, 3:5]
        con = np.zeros((P, 4), np.float32); con[:, [0, 1, 3]] = sums9[:, 5:8]
        g["dL_dmean2D"], g["dL_dconic"], g["dL_dopacity"] = m2.reshape(-1), con.reshape(-1), sums9[:, 8].copy()
        g["sums9"] = sums9
        return g


def check_pixels(r, got_color, got_T, got_last, alpha_margin=1e-4, T_margin=1e-4, rtol=1e-4, floor_T=1e-4, floor_C=1e-3, max_leaves=4096, exp_cond=0.0):
    """Every pixel of another implementation's forward output against the ADMISSIBLE blends of Rasterizer `r`'s last
    forward (gs_oracle.cpp, orc_check_pixels_f32): the nominal blend, or the blend with fragile decisions — pairs within
    alpha_margin / T_margin (relative) of the alpha = 1/255 / T = 1e-4 thresholds — taken the other way.
    exp_cond > 0 widens the comparison by the conditioning of the exponent: alpha_i may deviate by exp_cond * 2^-24 * m_i relative,
    m_i = (|a| dx^2 + |c| dy^2) / 2 + |b dx dy| (the magnitudes the power's three products cancel from), carried through the blend.
    Returns (status int32[N]: 0 nominal, 1 another admissible blend, 2 none, 3 undecided; leaves int32[N])."""
    a = r.args
    assert r.dt == np.float32
    N = a["W"] * a["H"]
    col = _c(got_color, np.float32).reshape(-1)
    gT = _c(got_T, np.float32).reshape(-1)
    gl = _c(got_last, np.uint32).reshape(-1)
    assert col.size == 3 * N and gT.size == N and gl.size == N
    status = np.zeros(N, np.int32)
    leaves = np.zeros(N, np.int32)
    lib().orc_check_pixels_f32(r.h, _p(a["bg"], f32p), _p(col, f32p), _p(gT, f32p), _p(gl, u32p), C.c_float(alpha_margin),
                               C.c_float(T_margin), C.c_float(rtol), C.c_float(floor_T), C.c_float(floor_C), C.c_int(max_leaves),
                               _p(status, i32p), _p(leaves, i32p), C.c_float(exp_cond))
    return status, leaves


def check_frame(r, frame, alpha_margin=1e-4, T_margin=1e-4, rtol=1e-4, floor_C=1e-3, max_leaves=4096, exp_cond=0.0):
    """An RGBA8 frame (uint32[N], imageFloatToInt layout) of the scene Rasterizer `r` last blended — Trainer::render's output — against
    the admissible blends of every pixel (gs_oracle.cpp, orc_check_frame_f32): accepted when some admissible blend's colour, within the pixel
    tolerance, quantises to the bytes the frame holds.  Returns (status int32[N]: 0 nominal, 1 another admissible blend, 2 none, 3 undecided; leaves)."""
    a = r.args
    assert r.dt == np.float32
    N = a["W"] * a["H"]
    fr = _c(frame, np.uint32).reshape(-1)
    assert fr.size == N
    status = np.zeros(N, np.int32)
    leaves = np.zeros(N, np.int32)
    lib().orc_check_frame_f32(r.h, _p(a["bg"], f32p), _p(fr, u32p), C.c_float(alpha_margin), C.c_float(T_margin), C.c_float(rtol), C.c_float(floor_C),
                              C.c_int(max_leaves), _p(status, i32p), _p(leaves, i32p), C.c_float(exp_cond))
    return status, leaves


def chain(r, sums9):
    """The per-splat half of the backward (linear in the nine pixel-stage sums) of Rasterizer `r`'s last forward on
    caller-supplied sums9[P, 9]; returns dL_dmean3D / dL_dcov3D / dL_dsh / dL_dscale / dL_drot (fp32)."""
    a = r.args
    P, M = r.P, a["M"]
    assert r.dt == np.float32
    sums9 = _c(sums9, np.float32).reshape(P, 9)
    g = dict(dL_dmean3D=np.zeros(3 * P, np.float32), dL_dcov3D=np.zeros(6 * P, np.float32), dL_dsh=np.zeros(3 * M * P, np.float32),
             dL_dscale=np.zeros(3 * P, np.float32), dL_drot=np.zeros(4 * P, np.float32))
    lib().orc_chain_f32(r.h, C.c_int(a["D"]), C.c_int(M), _p(a["means"], f32p), _p(a["shs"], f32p), _p(a["scales"], f32p),
                        C.c_float(a["mod"]), _p(a["rots"], f32p), _p(a["view"], f32p), _p(a["proj"], f32p), _p(a["campos"], f32p),
                        C.c_float(a["tanx"]), C.c_float(a["tany"]), _p(sums9, f32p), _p(g["dL_dmean3D"], f32p),
                        _p(g["dL_dcov3D"], f32p), _p(g["dL_dsh"], f32p), _p(g["dL_dscale"], f32p), _p(g["dL_drot"], f32p))
    return g


def image_float_to_int(src, w, h):
    src = _c(src, np.float32).reshape(-1)
    fb = np.zeros(w * h, np.uint32)
    lib().orc_image_float_to_int(_p(src, f32p), _p(fb, u32p), C.c_int(w), C.c_int(h))
    return fb


def image_int_to_loss(truth, rast, w, h):
    truth = _c(truth, np.uint32).reshape(-1)
    rast = _c(rast, np.float32).reshape(-1)
    loss = np.zeros(3 * w * h, np.float32)
    lib().orc_image_int_to_loss(_p(truth, u32p), _p(rast, f32p), _p(loss, f32p), C.c_int(w), C.c_int(h))
    return loss


def train_views(P, D, M, W, H, loc, sh, scale, opac, rot, views40, truths, samples, want_images=False):
    """Trainer::train's per-view loop (src/Trainer.cu:303-425).  Returns dict of var/avg buffers."""
    V = views40.shape[0]
    loc, sh, scale, opac, rot = (_c(x, np.float32).reshape(-1) for x in (loc, sh, scale, opac, rot))
    views40 = _c(views40, np.float32)
    truths = _c(truths, np.uint32).reshape(-1)
    o = dict(var=np.zeros(P, np.float32), loc=np.zeros(3 * P, np.float32), sh=np.zeros(3 * M * P, np.float32),
             scale=np.zeros(3 * P, np.float32), opac=np.zeros(P, np.float32), rot=np.zeros(4 * P, np.float32))
    nr = np.zeros(V, np.int32)
    imgs = np.zeros((V, 3, H, W), np.float32) if want_images else None
    lib().orc_train_views(C.c_int(P), C.c_int(D), C.c_int(M), C.c_int(W), C.c_int(H), C.c_int(V), _p(loc, f32p),
                          _p(sh, f32p), _p(scale, f32p), _p(opac, f32p), _p(rot, f32p), _p(views40, f32p),
                          _p(truths, u32p), C.c_float(samples), _p(o["var"], f32p), _p(o["loc"], f32p),
                          _p(o["sh"], f32p), _p(o["scale"], f32p), _p(o["opac"], f32p), _p(o["rot"], f32p),
                          _p(nr, i32p), _p(imgs, f32p))
    o["num_rendered"] = nr
    if want_images:
        o["images"] = imgs
    return o


def apply_sgd(loc, sh, scale, opac, rot, g, lrs, max_scale, M):
    """applyGradients (src/Trainer.cu:81-101), in place on float32 arrays."""
    P = opac.size
    lib().orc_apply_sgd(_p(loc, f32p), _p(sh, f32p), _p(scale, f32p), _p(opac, f32p), _p(rot, f32p),
                        _p(g["loc"], f32p), _p(g["sh"], f32p), _p(g["scale"], f32p), _p(g["opac"], f32p),
                        _p(g["rot"], f32p), *[C.c_float(x) for x in lrs], C.c_float(max_scale), C.c_int(M), C.c_int(P))


def apply_adam(loc, sh, scale, opac, rot, g, m, v, t, lrs, max_scale, b1, b2, eps, M):
    P = opac.size
    lib().orc_apply_adam(_p(loc, f32p), _p(sh, f32p), _p(scale, f32p), _p(opac, f32p), _p(rot, f32p),
                         _p(g["loc"], f32p), _p(g["sh"], f32p), _p(g["scale"], f32p), _p(g["opac"], f32p),
                         _p(g["rot"], f32p), _p(m, f32p), _p(v, f32p), C.c_int(t), *[C.c_float(x) for x in lrs],
                         C.c_float(max_scale), C.c_float(b1), C.c_float(b2), C.c_float(eps), C.c_int(M), C.c_int(P))


def densify(loc, sh, scale, opac, rot, count, capacity, M, var, grad_loc, hp, quat_xyzw=1, adam_m=None, adam_v=None):
    """Trainer::train densify block (src/Trainer.cu:437-542); arrays sized to capacity, in place.  adam_m / adam_v
    (optional, capacity-sized five-array layout) are carried along with their splats."""
    if adam_m is not None:
        lib().orc_densify_adam.restype = C.c_int
        return lib().orc_densify_adam(_p(loc, f32p), _p(sh, f32p), _p(scale, f32p), _p(opac, f32p), _p(rot, f32p),
                                      C.c_int(count), C.c_int(capacity), C.c_int(M), _p(_c(var, np.float32), f32p),
                                      _p(_c(grad_loc, np.float32), f32p), C.c_float(hp["cull_opacity"]),
                                      C.c_float(hp["cull_size"]), C.c_float(hp["densify_variance"]),
                                      C.c_float(hp["split_size"]), C.c_float(hp["split_distance"]),
                                      C.c_float(hp["split_scale"]), C.c_float(hp["clone_distance"]), C.c_int(quat_xyzw),
                                      _p(adam_m, f32p), _p(adam_v, f32p))
    return lib().orc_densify(_p(loc, f32p), _p(sh, f32p), _p(scale, f32p), _p(opac, f32p), _p(rot, f32p),
                             C.c_int(count), C.c_int(capacity), C.c_int(M), _p(_c(var, np.float32), f32p),
                             _p(_c(grad_loc, np.float32), f32p), C.c_float(hp["cull_opacity"]),
                             C.c_float(hp["cull_size"]), C.c_float(hp["densify_variance"]),
                             C.c_float(hp["split_size"]), C.c_float(hp["split_distance"]),
                             C.c_float(hp["split_scale"]), C.c_float(hp["clone_distance"]), C.c_int(quat_xyzw))


def fibonacci_sphere(count, distance):
    out = np.zeros((count, 3), np.float32)
    lib().orc_fibonacci_sphere(C.c_int(count), C.c_float(distance), _p(out, f32p))
    return out


def camera_view(loc, target=(0, 0, 0)):
    out = np.zeros(16, np.float32)
    lib().orc_camera_view(_p(_c(loc, np.float32), f32p), _p(_c(target, np.float32), f32p), _p(out, f32p))
    return out


def camera_proj(fov_deg_y, aspect):
    out = np.zeros(16, np.float32)
    lib().orc_camera_proj(C.c_float(fov_deg_y), C.c_float(aspect), _p(out, f32p))
    return out


def mat4_mul(a, b):
    out = np.zeros(16, np.float32)
    lib().orc_mat4_mul(_p(_c(a, np.float32), f32p), _p(_c(b, np.float32), f32p), _p(out, f32p))
    return out


def num_threads():
    return lib().orc_num_threads()
