"""ModelSplatsHost / ModelSplatsDevice — Python mirror of src/ModelSplatsHost.{h,cpp} and
src/ModelSplatsDevice.{h,cpp} over the C-ABI (same member names, argument meaning, error behaviour)."""
import ctypes as C

import numpy as np

from . import capi


class ModelSplatsHost:
    """src/ModelSplatsHost.h:8-39.  Arrays are allocated to `capacity`, `count` splats are valid."""

    def __init__(self, capacity, shDegree, shCoeffs):
        self.capacity, self.shDegree, self.shCoeffs = int(capacity), int(shDegree), int(shCoeffs)
        self.count = 0
        self.locations = np.zeros(self.capacity * 3, np.float32)
        self.shs = np.zeros(self.capacity * 3 * self.shCoeffs, np.float32)
        self.scales = np.zeros(self.capacity * 3, np.float32)
        self.opacities = np.zeros(self.capacity, np.float32)
        self.rotations = np.zeros(self.capacity * 4, np.float32)

    @classmethod
    def fromDevice(cls, device):
        """ModelSplatsHost(const ModelSplatsDevice&), src/ModelSplatsHost.cpp:16-26."""
        h = cls(device.capacity, device.shDegree, device.shCoeffs)
        h.count = device.count
        n, M = h.count, h.shCoeffs
        if n:
            loc, sh, sc, op, ro = (np.zeros(k, np.float32) for k in (3 * n, 3 * M * n, 3 * n, n, 4 * n))
            capi.check(capi.lib().gs_model_download(device.handle, *[a.ctypes.data_as(C.c_void_p) for a in (loc, sh, sc, op, ro)]))
            h.locations[:3 * n], h.shs[:3 * M * n], h.scales[:3 * n], h.opacities[:n], h.rotations[:4 * n] = loc, sh, sc, op, ro
        return h

    @classmethod
    def fromVectors(cls, locations, shs, scales, opacities, rotations):
        """The five-vector constructor, src/ModelSplatsHost.cpp:28-55 (capacity 1e6 grown x10,
        shDegree = (M-1)/3 integer quirk, dimension validation -> runtime_error)."""
        locations, shs, scales, opacities, rotations = (np.asarray(a, np.float32).reshape(-1)
                                                        for a in (locations, shs, scales, opacities, rotations))
        capacity = 1000000
        while capacity < locations.size // 3:
            capacity *= 10
        count = locations.size // 3
        if count == 0:
            raise RuntimeError("Inconsistent feature dimensions supplied when creating a host model!")
        shCoeffs = shs.size // (3 * count)
        shDegree = (shCoeffs - 1) // 3
        if (locations.size != count * 3 or shs.size != count * 3 * shCoeffs or scales.size != count * 3
                or opacities.size != count or rotations.size != count * 4 or shCoeffs < 1):
            raise RuntimeError("Inconsistent feature dimensions supplied when creating a host model!")
        h = cls(capacity, shDegree, shCoeffs)
        h.count = count
        h.locations[:3 * count], h.shs[:3 * shCoeffs * count], h.scales[:3 * count] = locations, shs, scales
        h.opacities[:count], h.rotations[:4 * count] = opacities, rotations
        return h

    def pushBack(self, location, sh, scale, opacity, rotation):
        """src/ModelSplatsHost.cpp:65-77.  `rotation` is the 4 floats as they are memcpy'd."""
        if self.count >= self.capacity:
            raise RuntimeError("Model ran out of capacity!")
        i, M = self.count, self.shCoeffs
        self.locations[3 * i:3 * i + 3] = location
        sh = list(sh)
        for k in range(3 * M):
            self.shs[3 * M * i + k] = sh[k]  # sh.at(i): IndexError where the reference throws out_of_range
        self.scales[3 * i:3 * i + 3] = scale
        self.opacities[i] = opacity
        self.rotations[4 * i:4 * i + 4] = rotation
        self.count += 1

    def copy(self, indexTo, indexFrom):
        """src/ModelSplatsHost.cpp:79-91."""
        if indexTo < 0 or indexTo >= self.count or indexFrom < 0 or indexFrom >= self.count:
            raise RuntimeError("Can't copy splat in model, incorrect bounds and/or no capacity!")
        M = self.shCoeffs
        self.locations[3 * indexTo:3 * indexTo + 3] = self.locations[3 * indexFrom:3 * indexFrom + 3]
        self.shs[3 * M * indexTo:3 * M * (indexTo + 1)] = self.shs[3 * M * indexFrom:3 * M * (indexFrom + 1)]
        self.scales[3 * indexTo:3 * indexTo + 3] = self.scales[3 * indexFrom:3 * indexFrom + 3]
        self.opacities[indexTo] = self.opacities[indexFrom]
        self.rotations[4 * indexTo:4 * indexTo + 4] = self.rotations[4 * indexFrom:4 * indexFrom + 4]


class ModelSplatsDevice:
    """src/ModelSplatsDevice.h:5-30 over a gs_model handle (device layout is SoA inside the library)."""

    def __init__(self, source=None, _handle=None, _owned=True):
        self._owned = _owned
        if _handle is not None:
            self.handle = C.c_void_p(_handle)
        elif isinstance(source, ModelSplatsHost):  # ModelSplatsDevice(const ModelSplatsHost&), .cpp:24-40
            h = source
            out = C.c_void_p()
            ptrs = [a.ctypes.data_as(C.c_void_p) for a in (h.locations, h.shs, h.scales, h.opacities, h.rotations)]
            capi.check(capi.lib().gs_model_create(h.capacity, h.shDegree, h.shCoeffs, h.count, *ptrs, C.byref(out)))
            self.handle = out
        elif isinstance(source, ModelSplatsDevice):  # copy constructor, .cpp:6-22
            out = C.c_void_p()
            capi.check(capi.lib().gs_model_clone(source.handle, C.byref(out)))
            self.handle = out
        else:
            raise TypeError("ModelSplatsDevice(ModelSplatsHost | ModelSplatsDevice)")

    def _info(self):
        v = [C.c_int() for _ in range(4)]
        capi.check(capi.lib().gs_model_info(self.handle, *[C.byref(x) for x in v]))
        return [x.value for x in v]

    capacity = property(lambda s: s._info()[0])
    shDegree = property(lambda s: s._info()[1])
    shCoeffs = property(lambda s: s._info()[2])
    count = property(lambda s: s._info()[3])

    def release(self):
        """Hand ownership to a trainer (gs_trainer_set_model)."""
        self._owned = False

    def __del__(self):
        try:
            if self._owned and self.handle:
                capi.lib().gs_model_destroy(self.handle)
        except Exception:
            pass
