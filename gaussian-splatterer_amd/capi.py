"""ctypes binding of include/gsplat.h (libgsplat_mi355.so).  No fallback: a missing library raises."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GSPLAT_MI355_LIB") or os.path.join(_HERE, "libgsplat_mi355.so")  # override: kernel-tuning builds only

GS_OK = 0
GS_ERR_INVALID_ARGUMENT, GS_ERR_HIP, GS_ERR_NO_TRUTH, GS_ERR_NO_DEVICE, GS_ERR_COLLECTIVE = -1, -2, -3, -9, -11
GS_UPDATE_SGD_CLAMP, GS_UPDATE_ADAM = 0, 1
GS_QUAT_WXYZ, GS_QUAT_XYZW = 0, 1
GS_COMM_ID_BYTES = 128
GS_STAGE_COUNT = 9

f32p = C.POINTER(C.c_float)
u32p = C.POINTER(C.c_uint32)


class GsError(RuntimeError):
    """What the C++ shim re-throws as std::runtime_error (the reference's error type)."""

    def __init__(self, status, message):
        super().__init__(f"[gs_status {status}] {message}")
        self.status = status


class gs_view(C.Structure):
    _fields_ = [("view", C.c_float * 16), ("projview", C.c_float * 16), ("campos", C.c_float * 3),
                ("tan_fovx", C.c_float), ("tan_fovy", C.c_float), ("bg", C.c_float * 3)]


class gs_hyper(C.Structure):
    _fields_ = [("lr_location", C.c_float), ("lr_sh", C.c_float), ("lr_scale", C.c_float), ("lr_opacity", C.c_float),
                ("lr_rotation", C.c_float), ("scale_max", C.c_float), ("cull_opacity", C.c_float),
                ("cull_size", C.c_float), ("densify_variance", C.c_float), ("split_size", C.c_float),
                ("split_distance", C.c_float), ("split_scale", C.c_float), ("clone_distance", C.c_float),
                ("update_rule", C.c_int), ("adam_beta1", C.c_float), ("adam_beta2", C.c_float), ("adam_eps", C.c_float),
                ("quat_layout", C.c_int)]


class gs_step_stats(C.Structure):
    _fields_ = [("count_before", C.c_int), ("count_after", C.c_int), ("views", C.c_int),
                ("num_rendered", C.c_longlong), ("max_tile_list", C.c_int), ("arena_regrows", C.c_int),
                ("loss", C.c_float)]


ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_size_t, C.c_void_p)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p)

# every symbol include/gsplat.h declares (tests/test_capi_symbols.py checks the library exports them all)
SYMBOLS = [
    "gs_last_error", "gs_status_string", "gs_version", "gs_device_count", "gs_set_option", "gs_debug_wave_reduce9", "gs_debug_counters", "gs_debug_hbm_copy_rate", "gs_debug_hbm_copy_form", "gs_device_malloc", "gs_device_free",
    "gs_memcpy_h2d", "gs_memcpy_d2h", "gs_memset_d", "gs_device_synchronize", "gs_model_create", "gs_model_clone",
    "gs_model_download", "gs_model_info", "gs_model_destroy", "gs_hyper_defaults", "gs_trainer_create",
    "gs_trainer_destroy", "gs_trainer_set_model", "gs_trainer_get_model", "gs_trainer_set_views", "gs_trainer_step",
    "gs_trainer_accumulate", "gs_trainer_grad_buffer", "gs_trainer_apply", "gs_trainer_set_allreduce",
    "gs_trainer_get_stream", "gs_trainer_synchronize", "gs_trainer_render", "gs_trainer_read_image",
    "gs_trainer_set_option", "gs_trainer_list_cut_stats", "gs_trainer_debug_list_totals", "gs_trainer_set_sharded_update", "gs_trainer_attach_comm_sharded", "gs_trainer_set_compact_exchange", "gs_trainer_attach_comm_compact", "gs_trainer_set_profiling", "gs_trainer_stage_times", "gs_stage_name", "gs_trainer_adam_state", "gs_trainer_set_adam_state",
    "gs_comm_unique_id", "gs_comm_create", "gs_comm_destroy", "gs_trainer_attach_comm", "gs_rasterize_forward",
    "gs_rasterize_backward", "gs_raster_chunk_field", "gs_image_float_to_int", "gs_image_int_to_loss",
]

_lib = None


def lib():
    """Load libgsplat_mi355.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` or "
                          f"`make -C gaussian-splatterer_amd/csrc` (there is no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    L.gs_last_error.restype = C.c_char_p
    L.gs_status_string.restype = C.c_char_p
    L.gs_status_string.argtypes = [C.c_int]
    vp, i, f = C.c_void_p, C.c_int, C.c_float
    L.gs_set_option.argtypes = [C.c_char_p, i]
    L.gs_debug_wave_reduce9.argtypes = [vp, vp]
    L.gs_debug_counters.argtypes = [vp, i]
    L.gs_debug_hbm_copy_rate.argtypes = [C.c_size_t, i, C.POINTER(C.c_double)]
    L.gs_device_malloc.argtypes = [C.POINTER(vp), C.c_size_t]
    L.gs_device_free.argtypes = [vp]
    L.gs_memcpy_h2d.argtypes = [vp, vp, C.c_size_t]
    L.gs_memcpy_d2h.argtypes = [vp, vp, C.c_size_t]
    L.gs_memset_d.argtypes = [vp, i, C.c_size_t]
    L.gs_model_create.argtypes = [i, i, i, i, vp, vp, vp, vp, vp, C.POINTER(vp)]
    L.gs_model_clone.argtypes = [vp, C.POINTER(vp)]
    L.gs_model_download.argtypes = [vp, vp, vp, vp, vp, vp]
    L.gs_model_info.argtypes = [vp, C.POINTER(i), C.POINTER(i), C.POINTER(i), C.POINTER(i)]
    L.gs_model_destroy.argtypes = [vp]
    L.gs_hyper_defaults.argtypes = [C.POINTER(gs_hyper)]
    L.gs_trainer_create.argtypes = [i, i, C.POINTER(vp)]
    L.gs_trainer_destroy.argtypes = [vp]
    L.gs_trainer_set_model.argtypes = [vp, vp]
    L.gs_trainer_get_model.argtypes = [vp]
    L.gs_trainer_get_model.restype = vp
    L.gs_trainer_set_views.argtypes = [vp, i, vp, vp, i, i]
    L.gs_trainer_set_option.argtypes = [vp, C.c_char_p, i]
    L.gs_trainer_list_cut_stats.argtypes = [vp, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
    L.gs_trainer_debug_list_totals.argtypes = [vp, C.POINTER(C.c_longlong)]
    L.gs_trainer_step.argtypes = [vp, C.POINTER(gs_hyper), i, C.POINTER(gs_step_stats)]
    L.gs_trainer_accumulate.argtypes = [vp, C.POINTER(gs_step_stats)]
    L.gs_trainer_grad_buffer.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.gs_trainer_apply.argtypes = [vp, C.POINTER(gs_hyper), i, C.POINTER(gs_step_stats)]
    L.gs_trainer_set_allreduce.argtypes = [vp, vp, vp]
    L.gs_trainer_adam_state.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(i)]
    L.gs_trainer_set_adam_state.argtypes = [vp, vp, vp, C.c_size_t, i, i]
    L.gs_trainer_get_stream.argtypes = [vp, C.POINTER(vp)]
    L.gs_trainer_synchronize.argtypes = [vp]
    L.gs_trainer_render.argtypes = [vp, vp, i, i, i, f, C.POINTER(gs_view)]
    L.gs_trainer_read_image.argtypes = [vp, i, vp]
    L.gs_trainer_set_profiling.argtypes = [vp, i]
    L.gs_trainer_stage_times.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_longlong)]
    L.gs_stage_name.argtypes = [i]
    L.gs_stage_name.restype = C.c_char_p
    L.gs_comm_unique_id.argtypes = [vp]
    L.gs_comm_create.argtypes = [vp, i, i, C.POINTER(vp)]
    L.gs_comm_destroy.argtypes = [vp]
    L.gs_trainer_attach_comm.argtypes = [vp, vp]
    L.gs_trainer_attach_comm_sharded.argtypes = [vp, vp]
    L.gs_trainer_set_sharded_update.argtypes = [vp, vp, vp, vp, i, i]
    L.gs_trainer_set_compact_exchange.argtypes = [vp, vp, vp, vp, i, i, i, vp]
    L.gs_trainer_attach_comm_compact.argtypes = [vp, vp, vp, i, vp]
    L.gs_rasterize_forward.argtypes = [ALLOC_FN, vp, ALLOC_FN, vp, ALLOC_FN, vp, i, i, i, vp, i, i, vp, vp, vp, vp, vp, f,
                                       vp, vp, vp, vp, vp, f, f, i, vp, vp, i, C.POINTER(i)]
    L.gs_rasterize_backward.argtypes = [i, i, i, i, vp, i, i, vp, vp, vp, vp, f, vp, vp, vp, vp, vp, f, f, vp, vp, vp, vp,
                                        vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i]
    L.gs_raster_chunk_field.argtypes = [C.c_char_p, C.c_char_p, i, i, i, i, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.gs_image_float_to_int.argtypes = [vp, vp, i, i]
    L.gs_image_int_to_loss.argtypes = [vp, vp, vp, i, i]
    _lib = L
    return L


def check(status):
    if status != GS_OK:
        raise GsError(status, lib().gs_last_error().decode(errors="replace") or lib().gs_status_string(status).decode())
    return status


def last_error():
    """The library's thread-local message of the last failing call (gs_last_error)."""
    return lib().gs_last_error().decode(errors="replace")


def hyper_defaults():
    h = gs_hyper()
    check(lib().gs_hyper_defaults(C.byref(h)))
    return h


def view_from_block(block40):
    v = gs_view()
    C.memmove(C.byref(v), np.ascontiguousarray(block40, np.float32).ctypes.data, 160)
    return v


class DeviceBuffer:
    """Raw device allocation through gs_device_malloc (for the device-pointer entry points)."""

    def __init__(self, nbytes):
        self.ptr = C.c_void_p()
        self.nbytes = int(nbytes)
        check(lib().gs_device_malloc(C.byref(self.ptr), self.nbytes))

    @classmethod
    def from_numpy(cls, a):
        a = np.ascontiguousarray(a)
        b = cls(max(a.nbytes, 4))
        if a.nbytes:
            check(lib().gs_memcpy_h2d(b.ptr, a.ctypes.data_as(C.c_void_p), a.nbytes))
        return b

    def to_numpy(self, dtype, count=None, offset=0):
        dtype = np.dtype(dtype)
        n = (self.nbytes - offset) // dtype.itemsize if count is None else count
        out = np.empty(n, dtype)
        if n:
            check(lib().gs_memcpy_d2h(out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr.value + offset), n * dtype.itemsize))
        return out

    def zero(self):
        check(lib().gs_memset_d(self.ptr, 0, self.nbytes))

    def free(self):
        if self.ptr:
            lib().gs_device_free(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
