"""Host-side mirror of the reference's Camera (src/Camera.h:9-26, src/Camera.cpp:9-86) and of the
per-view parameter block Trainer::train builds (src/Trainer.cu:311-326,355-356).

All matrices are glm column-major float32[16] (m[col*4+row]), exactly what glm::value_ptr hands
to cudaMemcpy in the reference.  A "view" is one (camera, background) pass: V = 2 * #cameras,
white backgrounds first, then black (src/Trainer.cu:311-314).
"""
import math

import numpy as np

F = np.float32
VIEW_FLOATS = 40  # struct gs_view: view[16] projview[16] campos[3] tan_fovx tan_fovy bg[3]


def fibonacci_sphere(count, distance):
    """getFibonacciSphere, src/Camera.cpp:9-27."""
    golden = (F(1.0) + np.sqrt(F(5.0))) / F(2.0)
    step = F(2.0) * F(math.pi) * golden
    out = np.zeros((count, 3), F)
    for i in range(count):
        t = F(i) / F(count)
        a1 = np.arccos(F(1.0) - F(2.0) * t)
        a2 = step * F(i)
        out[i] = (np.sin(a1) * np.cos(a2) * F(distance), np.sin(a1) * np.sin(a2) * F(distance), np.cos(a1) * F(distance))
    return out.astype(F)


def _norm(v):
    return (v * (F(1.0) / np.sqrt(np.dot(v, v).astype(F)))).astype(F)


def look_at_neg(loc, target=(0.0, 0.0, 0.0)):
    """Camera::getView, src/Camera.cpp:79-82: -glm::lookAt(location, target, +Y)."""
    eye = np.asarray(loc, F)
    center = np.asarray(target, F)
    up = np.array([0, 1, 0], F)
    f = _norm(center - eye)
    s = _norm(np.cross(f, up).astype(F))
    u = np.cross(s, f).astype(F)
    m = np.zeros(16, F)
    m[15] = 1
    m[0], m[4], m[8] = s
    m[1], m[5], m[9] = u
    m[2], m[6], m[10] = -f
    m[12], m[13], m[14] = -np.dot(s, eye), -np.dot(u, eye), np.dot(f, eye)
    return (-m).astype(F)


def perspective(fov_deg_y, aspect, z_near=0.1, z_far=100.0):
    """Camera::getProjection, src/Camera.cpp:84-86: glm::perspective(radians(fovY), aspect, 0.1, 100)."""
    fovy = F(fov_deg_y) * F(0.01745329251994329576923690768489)
    t = F(math.tan(float(fovy) / 2.0))
    zn, zf = F(z_near), F(z_far)
    m = np.zeros(16, F)
    m[0] = F(1.0) / (F(aspect) * t)
    m[5] = F(1.0) / t
    m[10] = -(zf + zn) / (zf - zn)
    m[11] = F(-1.0)
    m[14] = -(F(2.0) * zf * zn) / (zf - zn)
    return m


def mat4_mul(a, b):
    """glm mat4 * mat4 on column-major arrays."""
    A = np.asarray(a, F).reshape(4, 4).T  # A[row, col]
    B = np.asarray(b, F).reshape(4, 4).T
    return (A @ B).T.reshape(16).astype(F)


class Camera:
    """src/Camera.h:9-26."""

    def __init__(self, location, target=(0.0, 0.0, 0.0), fov_deg_y=60.0):
        self.location = np.asarray(location, F)
        self.target = np.asarray(target, F)
        self.fovDegY = float(fov_deg_y)

    def getView(self):
        return look_at_neg(self.location, self.target)

    def getProjection(self, aspect):
        return perspective(self.fovDegY, aspect)


def get_cameras(count, distance=10.0, fov_deg=60.0):
    """Camera::getCameras with one sphere, rotX = rotY = 0 (src/Camera.cpp:33-47, Project.h:14-19)."""
    return [Camera(p, (0, 0, 0), fov_deg) for p in fibonacci_sphere(count, distance)]


def _angle_axis_mat3(angle_rad, axis):
    """(glm::mat4)glm::angleAxis(angle, axis): quaternion (cos a/2, axis sin a/2) -> rotation matrix, row-major 3x3."""
    a = F(angle_rad)
    s, c = np.sin(a * F(0.5)), np.cos(a * F(0.5))
    x, y, z = (F(axis[0]) * s, F(axis[1]) * s, F(axis[2]) * s)
    w = c
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]], F)


def get_cameras_project(project):
    """Camera::getCameras(project), src/Camera.cpp:33-58: two Fibonacci spheres, each rotated by
    angleAxis(radians(rotX), +Y) * angleAxis(radians(rotY), +X); all cameras look at the origin."""
    out = []
    for sp in (project.sphere1, project.sphere2):
        rot = (_angle_axis_mat3(math.radians(sp.rotX), (0, 1, 0)) @ _angle_axis_mat3(math.radians(sp.rotY), (1, 0, 0))).astype(F)
        for p in fibonacci_sphere(sp.count, sp.distance):
            out.append(Camera((rot @ p).astype(F), (0, 0, 0), sp.fovDeg))
    return out


def get_preview_camera(project):
    """Camera::getPreviewCamera, src/Camera.cpp:60-74 — the camera every Trainer::render caller of the reference builds
    (src/ui/UiPanelViewOutput.cpp:52-60, src/ui/tools/UiPanelToolsView.cpp:250): one of the truth cameras
    (project.previewTruth; an index past the end raises like std::vector::at), or the free camera
    (0, 0, -previewFreeDistance) turned by angleAxis(radians(previewFreeRotY) + orbit, +Y) * angleAxis(radians(previewFreeRotX), +X).
    The reference adds previewTimer * previewFreeOrbitSpeed to the angle AFTER converting previewFreeRotY to radians
    (:68-69: the term it calls degRotOrbit is used as radians); reproduced as is."""
    if project.previewTruth:
        cams = get_cameras_project(project)
        if not 0 <= project.previewTruthIndex < len(cams):
            raise IndexError("vector::at: previewTruthIndex %d of %d cameras" % (project.previewTruthIndex, len(cams)))
        return cams[project.previewTruthIndex]
    orbit = F(project.previewTimer) * F(project.previewFreeOrbitSpeed) if project.previewFreeOrbit else F(0.0)
    rot = (_angle_axis_mat3(F(math.radians(project.previewFreeRotY)) + orbit, (0, 1, 0)) @ _angle_axis_mat3(math.radians(project.previewFreeRotX), (1, 0, 0))).astype(F)
    loc = (rot @ np.array([0.0, 0.0, -project.previewFreeDistance], F)).astype(F)
    return Camera(loc, (0, 0, 0), project.previewFreeFovDeg)


def get_cameras_count(project):
    """Camera::getCamerasCount, src/Camera.cpp:29-31."""
    return project.sphere1.count + project.sphere2.count


def view_block(cam, width, height, white):
    """The 40-float gs_view the reference computes per pass (src/Trainer.cu:317-326,355-356)."""
    v = np.zeros(VIEW_FLOATS, F)
    view = cam.getView()
    v[0:16] = view
    v[16:32] = mat4_mul(cam.getProjection(F(width) / F(height)), view)
    v[32:35] = cam.location
    t = F(math.tan(math.radians(cam.fovDegY) * 0.5))
    v[35] = t
    v[36] = t
    v[37:40] = 1.0 if white else 0.0
    return v


def train_views(cameras, width, height):
    """All 2*C passes of one training iteration in the reference's order: C white, then C black."""
    C = len(cameras)
    out = np.zeros((2 * C, VIEW_FLOATS), F)
    for i in range(2 * C):
        out[i] = view_block(cameras[i % C], width, height, white=(i < C))
    return out
