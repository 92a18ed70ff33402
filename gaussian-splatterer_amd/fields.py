"""Splat-field initialisers — the step BEFORE the training path: UiFrame::initFieldGrid /
initFieldMono / initFieldModel (src/ui/UiFrame.cpp:137-264) as functions returning a ModelSplatsHost.

Quaternion storage: the reference pushes `glm::angleAxis(angle, axis)` through a raw memcpy of the glm::quat
object (src/ModelSplatsHost.cpp:74).  With glm's default member order {x, y, z, w} the identity is stored as
[0, 0, 0, 1]; `quat_xyzw=False` models GLM_FORCE_QUAT_DATA_WXYZ ([1, 0, 0, 0]).  The rasterizer reads element
0 as the real part either way — for the isotropic grid / mono fields both give the same covariance."""
import math

import numpy as np

from .model import ModelSplatsHost

SPLATS_LIMIT = 1000000   # src/Config.h:17
SPLATS_SH_DEGREE = 1     # src/Config.h:19
SPLATS_SH_COEF = 4       # src/Config.h:20
F = np.float32


def _angle_axis_memory(angle, axis, quat_xyzw=True):
    """glm::angleAxis(angle, axis) as the 4 floats a memcpy of the glm::quat yields (axis is NOT normalised by glm)."""
    a = F(angle)
    s = F(math.sin(float(a) * 0.5))
    w = F(math.cos(float(a) * 0.5))
    x, y, z = F(axis[0]) * s, F(axis[1]) * s, F(axis[2]) * s
    return [x, y, z, w] if quat_xyzw else [w, x, y, z]


def initFieldGrid(capacity=SPLATS_LIMIT, shDegree=SPLATS_SH_DEGREE, shCoeffs=SPLATS_SH_COEF, quat_xyzw=True):
    """src/ui/UiFrame.cpp:137-160: 17^3 splats on [-4, 4] step 0.5, scale 0.05, opacity 1, SH 0 (grey 0.5)."""
    h = ModelSplatsHost(capacity, shDegree, shCoeffs)
    dim, step = F(4.0), F(0.5)
    shs = [0.0] * (3 * shCoeffs)
    rot = _angle_axis_memory(0.0, (0.0, 1.0, 0.0), quat_xyzw)
    sc = [step * F(0.1)] * 3
    x = -dim
    while x <= dim:            # float accumulation exactly like the reference's `for(float x = -dim; x <= dim; x += step)`
        y = -dim
        while y <= dim:
            z = -dim
            while z <= dim:
                h.pushBack([x, y, z], shs, sc, 1.0, rot)
                z = F(z + step)
            y = F(y + step)
        x = F(x + step)
    return h


def initFieldMono(capacity=SPLATS_LIMIT, shDegree=SPLATS_SH_DEGREE, shCoeffs=SPLATS_SH_COEF, quat_xyzw=True):
    """src/ui/UiFrame.cpp:162-176: one giant splat (scale 0.3) at the origin."""
    h = ModelSplatsHost(capacity, shDegree, shCoeffs)
    h.pushBack([0.0, 0.0, 0.0], [0.0] * (3 * shCoeffs), [0.3, 0.3, 0.3], 1.0, _angle_axis_memory(0.0, (0.0, 1.0, 0.0), quat_xyzw))
    return h


def parse_obj_triangles(text):
    """The reference's own OBJ subset (src/ui/UiFrame.cpp:185-232): `v`, `vt` (ignored), `f` with 3 or 4
    `v/vt/vn` vertices (quads are fanned); any other face size raises like the reference."""
    vertices, triangles = [], []
    for line in text.splitlines():
        parts = line.split()
        if not parts:
            continue
        if parts[0] == "v":
            vertices.append([float(parts[1]), float(parts[2]), float(parts[3])])
        elif parts[0] == "f":
            idx = []
            for tok in parts[1:]:
                head = tok.split("/")[0]
                v = 0
                for ch in head:
                    v = v * 10 + (ord(ch) - ord("0"))
                idx.append(v)
            if len(idx) == 4:
                triangles.append((idx[0] - 1, idx[1] - 1, idx[2] - 1))
                triangles.append((idx[0] - 1, idx[2] - 1, idx[3] - 1))
            elif len(idx) == 3:
                triangles.append((idx[0] - 1, idx[1] - 1, idx[2] - 1))
            else:
                raise RuntimeError("Unexpected vertex count in face list!" + str(len(idx)))
    return np.asarray(vertices, F).reshape(-1, 3), triangles


def initFieldModel(obj_text, capacity=SPLATS_LIMIT, shDegree=SPLATS_SH_DEGREE, shCoeffs=SPLATS_SH_COEF, quat_xyzw=True):
    """src/ui/UiFrame.cpp:178-264: one thin splat per triangle, centred on it, scaled by two edge lengths x 0.2
    (thickness 0.005 x 0.2) and rotated from +Z onto the triangle normal with angleAxis(acos(n.z), cross(+Z, n))."""
    vertices, triangles = parse_obj_triangles(obj_text)
    h = ModelSplatsHost(capacity, shDegree, shCoeffs)
    shs = [0.0] * (3 * shCoeffs)
    up = np.array([0, 0, 1], F)
    for a, b, c in triangles:
        v0, v1, v2 = vertices[a], vertices[b], vertices[c]
        location = ((v0 + v1 + v2) / F(3.0)).astype(F)
        scale = (np.array([np.linalg.norm(v1 - v0), np.linalg.norm(v2 - v0), 0.005], F) * F(0.2)).astype(F)
        n = np.cross(v1 - v0, v2 - v0).astype(F)
        n = (n * (F(1.0) / np.sqrt(np.dot(n, n).astype(F)))).astype(F)
        axis = np.cross(up, n).astype(F)
        angle = F(math.acos(float(np.dot(up, n))))
        h.pushBack(location, shs, scale, 1.0, _angle_axis_memory(angle, axis, quat_xyzw))
    return h
