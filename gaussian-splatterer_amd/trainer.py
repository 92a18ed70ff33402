"""Trainer / Project — Python mirror of src/Trainer.cuh:10-75 and the run-time hyper-parameters of
src/Project.h:24-45 over the C-ABI.  Same public surface: .model, .truthFrameBuffersW/B,
.truthCameras, render(), train(project, densify); captureTruths() takes the truth images as input
because the reference's OptiX renderer is out of scope."""
import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import camera as cam
from . import capi
from .model import ModelSplatsDevice


@dataclass
class CameraSphere:
    """Project::CameraSphere, src/Project.h:14-22."""
    count: int = 16
    distance: float = 10.0
    fovDeg: float = 60.0
    rotX: float = 0.0
    rotY: float = 0.0


@dataclass
class Project:
    """src/Project.h:6-75: every serialised field, same names and defaults (JSON keys = member names)."""
    perspective: str = ""
    pathModel: str = ""
    pathTextureDiffuse: str = ""
    sphere1: CameraSphere = field(default_factory=CameraSphere)
    sphere2: CameraSphere = field(default_factory=CameraSphere)
    rtSamples: int = 100
    lrLocation: float = 0.00005
    lrSh: float = 0.0001
    lrScale: float = 0.00002
    lrOpacity: float = 0.0001
    lrRotation: float = 0.000025
    paramScaleMax: float = 0.3
    paramCullOpacity: float = 0.005
    paramCullSize: float = 0.004
    paramDensifyVariance: float = 2.0
    paramSplitSize: float = 0.04
    paramSplitDistance: float = 1.5
    paramSplitScale: float = 0.8
    paramCloneDistance: float = 1.6
    iterations: int = 0
    intervalCapture: int = 50
    intervalDensify: int = 200
    previewTimer: float = 0.0
    previewRtSamples: int = 50
    previewSplatScale: float = 1.0
    previewTruth: bool = False
    previewTruthIndex: int = 0
    previewFreeOrbit: bool = True
    previewFreeOrbitSpeed: float = 0.5
    previewFreeDistance: float = 10.0
    previewFreeFovDeg: float = 60.0
    previewFreeRotX: float = 25.0
    previewFreeRotY: float = 0.0
    renderResX: int = 2048
    renderResY: int = 2048
    # build-side extensions, NOT serialised: update rule and Adam constants
    updateRule: int = capi.GS_UPDATE_SGD_CLAMP
    adamBeta1: float = 0.9
    adamBeta2: float = 0.999
    adamEps: float = 1e-15
    quatLayout: int = capi.GS_QUAT_XYZW

    @classmethod
    def initProject(cls):
        """UiFrame::initProject, src/ui/UiFrame.cpp:129-134: a fresh project has an empty second sphere."""
        p = cls()
        p.sphere2.count = 0
        p.sphere2.fovDeg = 30.0
        return p

    def hyper(self):
        h = capi.gs_hyper()
        h.lr_location, h.lr_sh, h.lr_scale, h.lr_opacity, h.lr_rotation = (self.lrLocation, self.lrSh, self.lrScale,
                                                                            self.lrOpacity, self.lrRotation)
        h.scale_max = self.paramScaleMax
        h.cull_opacity, h.cull_size, h.densify_variance = self.paramCullOpacity, self.paramCullSize, self.paramDensifyVariance
        h.split_size, h.split_distance, h.split_scale, h.clone_distance = (self.paramSplitSize, self.paramSplitDistance,
                                                                            self.paramSplitScale, self.paramCloneDistance)
        h.update_rule = self.updateRule
        h.adam_beta1, h.adam_beta2, h.adam_eps = self.adamBeta1, self.adamBeta2, self.adamEps
        h.quat_layout = self.quatLayout
        return h


class Trainer:
    def __init__(self, width=1024, height=1024):
        self.width, self.height = int(width), int(height)
        self.handle = C.c_void_p()
        capi.check(capi.lib().gs_trainer_create(self.width, self.height, C.byref(self.handle)))
        self.truthFrameBuffersW = []  # host uint32 images (the reference holds device pointers)
        self.truthFrameBuffersB = []
        self.truthCameras = []
        self._views_dirty = True
        self._rank, self._world = 0, 1
        self._keepalive = []

    # ---- public member `model` (src/Trainer.cuh:50): assignment mirrors `delete model; model = new ...`
    @property
    def model(self):
        return ModelSplatsDevice(_handle=capi.lib().gs_trainer_get_model(self.handle), _owned=False)

    @model.setter
    def model(self, device_model):
        capi.check(capi.lib().gs_trainer_set_model(self.handle, device_model.handle))
        device_model.release()

    def captureTruths(self, cameras, framesW, framesB, view_blocks=None):
        """Replaces Trainer::captureTruths (src/Trainer.cu:218-250): the caller supplies, per camera, the
        white- and black-background RGBA8 truth images (width*height uint32 each).  view_blocks (optional,
        float32[2C, 40]): the pass parameters to use instead of deriving them from the cameras (C white, then C black)."""
        assert len(cameras) == len(framesW) == len(framesB)
        self._view_blocks = None if view_blocks is None else np.ascontiguousarray(view_blocks, np.float32).reshape(2 * len(cameras), 40)
        self.truthCameras = list(cameras)
        self.truthFrameBuffersW = [np.ascontiguousarray(f, np.uint32).reshape(-1) for f in framesW]
        self.truthFrameBuffersB = [np.ascontiguousarray(f, np.uint32).reshape(-1) for f in framesB]
        self._views_dirty = True

    def shard(self, rank, world):
        """Data-parallel view sharding: camera c of the iteration goes to rank c % world with both passes (dist.shard_views)."""
        self._rank, self._world = int(rank), int(world)
        self._views_dirty = True

    def _upload_views(self):
        Cn = len(self.truthCameras)
        blocks = cam.train_views(self.truthCameras, self.width, self.height) if Cn else np.zeros((0, 40), np.float32)
        if Cn and getattr(self, "_view_blocks", None) is not None:
            blocks = self._view_blocks
        total = 2 * Cn
        from .dist import shard_views
        mine = shard_views(total, self._rank, self._world)
        views = (capi.gs_view * max(len(mine), 1))()
        ptrs = (C.c_void_p * max(len(mine), 1))()
        for k, v in enumerate(mine):
            views[k] = capi.view_from_block(blocks[v])
            img = self.truthFrameBuffersW[v] if v < Cn else self.truthFrameBuffersB[v - Cn]
            assert img.size == self.width * self.height
            ptrs[k] = img.ctypes.data
        capi.check(capi.lib().gs_trainer_set_views(self.handle, len(mine), views, ptrs, 0, max(total, len(mine))))
        self._views_dirty = False
        self.local_views = mine

    def train(self, project, densify=False, stats=False):
        """Trainer::train(Project&, bool densify), src/Trainer.cu:252-543."""
        if not self.truthFrameBuffersW:
            raise RuntimeError("Can't run training iteration, no truth data available!")
        project.iterations += 1
        if self._views_dirty:
            self._upload_views()
        h = project.hyper()
        st = capi.gs_step_stats()
        capi.check(capi.lib().gs_trainer_step(self.handle, C.byref(h), int(bool(densify)), C.byref(st) if stats else None))
        return st if stats else None

    def accumulate(self, stats=False):
        """First half of train() (src/Trainer.cu:303-425): every pass's forward, loss, backward and gradient averaging.
        The averaged gradients are complete in grad_buffer() on return; the data-parallel driver reduces them here."""
        if not self.truthFrameBuffersW:
            raise RuntimeError("Can't run training iteration, no truth data available!")
        if self._views_dirty:
            self._upload_views()
        st = capi.gs_step_stats()
        capi.check(capi.lib().gs_trainer_accumulate(self.handle, C.byref(st) if stats else None))
        return st if stats else None

    def apply(self, project, densify=False, stats=False):
        """Second half of train() (src/Trainer.cu:427-542): the update and, when asked, densify/prune."""
        project.iterations += 1
        h = project.hyper()
        st = capi.gs_step_stats()
        capi.check(capi.lib().gs_trainer_apply(self.handle, C.byref(h), int(bool(densify)), C.byref(st) if stats else None))
        return st if stats else None

    def render(self, sizeX, sizeY, splatScale, camera, background=(0.0, 0.0, 0.0)):
        """Trainer::render, src/Trainer.cu:148-216; returns the RGBA8 framebuffer as uint32[sizeY, sizeX]."""
        import math
        blk = cam.view_block(camera, sizeX, sizeY, white=False)
        blk[35] = np.float32(math.tan(math.radians(sizeX * camera.fovDegY / sizeY) * 0.5))  # the reference's tan_fovx quirk, :196
        blk[37:40] = background
        v = capi.view_from_block(blk)
        fb = np.zeros(sizeX * sizeY, np.uint32)
        capi.check(capi.lib().gs_trainer_render(self.handle, fb.ctypes.data_as(C.c_void_p), 0, sizeX, sizeY,
                                                C.c_float(splatScale), C.byref(v)))
        return fb.reshape(sizeY, sizeX)

    def set_option(self, name, value):
        """gs_trainer_set_option: this trainer's switches (include/gsplat.h), e.g. "fuse_camera_passes", "sh_fp16"."""
        capi.check(capi.lib().gs_trainer_set_option(self.handle, name.encode(), int(value)))

    def list_cut_stats(self):
        """gs_trainer_list_cut_stats: (accumulate attempts that ran with depth-cut tile lists, those of them that were replayed uncut)."""
        a, b = C.c_longlong(), C.c_longlong()
        capi.check(capi.lib().gs_trainer_list_cut_stats(self.handle, C.byref(a), C.byref(b)))
        return a.value, b.value

    def synchronize(self):
        capi.check(capi.lib().gs_trainer_synchronize(self.handle))

    def read_image(self, local_view):
        out = np.zeros((3, self.height, self.width), np.float32)
        capi.check(capi.lib().gs_trainer_read_image(self.handle, local_view, out.ctypes.data_as(C.c_void_p)))
        return out

    def adam_state(self):
        """gs_trainer_adam_state: (moment1, moment2, steps) as host arrays [(11 + 3M) * plane stride] — or (None, None, 0)
        before the first Adam step.  Together with the model this is what a checkpoint of an Adam run holds."""
        m, v, n, steps = C.c_void_p(), C.c_void_p(), C.c_size_t(), C.c_int()
        capi.check(capi.lib().gs_trainer_adam_state(self.handle, C.byref(m), C.byref(v), C.byref(n), C.byref(steps)))
        if not m.value:
            return None, None, 0
        out = []
        for p in (m, v):
            a = np.empty(n.value, np.float32)
            capi.check(capi.lib().gs_memcpy_d2h(a.ctypes.data_as(C.c_void_p), p, n.value * 4))
            out.append(a)
        return out[0], out[1], steps.value

    def set_adam_state(self, moment1, moment2, steps):
        """gs_trainer_set_adam_state: resume an Adam run (call after assigning .model, which resets the optimizer state)."""
        if moment1 is None:
            capi.check(capi.lib().gs_trainer_set_adam_state(self.handle, None, None, 0, 0, 0))
            return
        m1 = np.ascontiguousarray(moment1, np.float32).reshape(-1)
        m2 = np.ascontiguousarray(moment2, np.float32).reshape(-1)
        assert m1.size == m2.size
        capi.check(capi.lib().gs_trainer_set_adam_state(self.handle, m1.ctypes.data_as(C.c_void_p), m2.ctypes.data_as(C.c_void_p), m1.size, int(steps), 0))

    def grad_buffer(self):
        p, n = C.c_void_p(), C.c_size_t()
        capi.check(capi.lib().gs_trainer_grad_buffer(self.handle, C.byref(p), C.byref(n)))
        return p.value, n.value

    def close(self):
        if self.handle:
            capi.lib().gs_trainer_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
