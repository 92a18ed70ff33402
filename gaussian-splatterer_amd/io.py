"""On-disk formats either side of the training path:
  * `.gobj` — the reference's text splat format, UiFrame::saveSplats / loadSplats (src/ui/UiFrame.cpp:333-358,
    :373-450): five lines per splat `v x y z`, `sh f0 f1 ...`, `s x y z`, `a opacity`, `r q0 q1 q2 q3`, floats at
    the C++ ostream default (6 significant digits, %g) — lossy, like the reference.
  * `settings.json` — Project via nlohmann's NLOHMANN_DEFINE_TYPE_INTRUSIVE_WITH_DEFAULT (src/Project.h:64-73):
    keys are the member names, missing keys keep their defaults."""
import dataclasses
import json

import numpy as np

from .model import ModelSplatsHost
from .trainer import CameraSphere, Project


def _g(x):
    return "%g" % float(np.float32(x))


def saveSplats(path, model):
    """UiFrame::saveSplats, src/ui/UiFrame.cpp:333-358.  `model` is a ModelSplatsHost (or anything with the same fields)."""
    M = model.shCoeffs
    with open(path, "w") as f:
        for i in range(model.count):
            f.write("v " + " ".join(_g(v) for v in model.locations[3 * i:3 * i + 3]) + "\n")
            f.write("sh" + "".join(" " + _g(v) for v in model.shs[3 * M * i:3 * M * (i + 1)]) + "\n")
            f.write("s " + " ".join(_g(v) for v in model.scales[3 * i:3 * i + 3]) + "\n")
            f.write("a " + _g(model.opacities[i]) + "\n")
            f.write("r " + " ".join(_g(v) for v in model.rotations[4 * i:4 * i + 4]) + "\n")


_STREAM_FLOAT = None


def _stream_floats(text, width=None):
    """Numbers the way successive `iss >> x` (float) read them: decimal forms only ("nan", "inf", hex are not numbers to a
    stream), stopping at the first token that is none.  width: exactly that many values, 0.0 once extraction has failed
    (the reference pushes 3 / 1 / 4 values per v, s / a / r line whatever the line holds, src/ui/UiFrame.cpp:404-434)."""
    global _STREAM_FLOAT
    import re
    if _STREAM_FLOAT is None:
        _STREAM_FLOAT = re.compile(r"\s*([+-]?(?:\d+\.?\d*|\.\d+)(?:[eE][+-]?\d+)?)")
    out, pos = [], 0
    while width is None or len(out) < width:
        m = _STREAM_FLOAT.match(text, pos)
        if not m:
            break
        out.append(float(np.float32(m.group(1))))
        pos = m.end()
    if width is not None:
        out += [0.0] * (width - len(out))
    return out


def loadSplats(path):
    """UiFrame::loadSplats, src/ui/UiFrame.cpp:373-450 -> ModelSplatsHost (five-vector constructor semantics:
    capacity 1e6 grown x10, shDegree = (M-1)/3; inconsistent SH counts raise "Inconsistent SH degree!")."""
    loc, shs, sc, op, rot = [], [], [], [], []
    sh_coeffs = None
    with open(path) as f:
        for line in f:
            parts = line.split(None, 1)
            if not parts:
                continue
            p, rest = parts[0], (parts[1] if len(parts) > 1 else "")
            if p == "v":
                loc += _stream_floats(rest, 3)
            elif p == "sh":
                got = _stream_floats(rest)   # `while (iss >> x)`: stops at the first token that is not a number
                shs += got
                if sh_coeffs is None:
                    sh_coeffs = len(got)
                elif sh_coeffs != len(got):
                    raise RuntimeError("Inconsistent SH degree!")
            elif p == "s":
                sc += _stream_floats(rest, 3)
            elif p == "a":
                op += _stream_floats(rest, 1)
            elif p == "r":
                rot += _stream_floats(rest, 4)
    return ModelSplatsHost.fromVectors(loc, shs, sc, op, rot)


_SERIALISED = ["perspective", "pathModel", "pathTextureDiffuse", "sphere1", "sphere2", "rtSamples", "lrLocation", "lrSh", "lrScale",
               "lrOpacity", "lrRotation", "paramScaleMax", "paramCullOpacity", "paramCullSize", "paramDensifyVariance",
               "paramSplitSize", "paramSplitDistance", "paramSplitScale", "paramCloneDistance", "iterations", "intervalCapture",
               "intervalDensify", "previewTimer", "previewRtSamples", "previewSplatScale", "previewTruth", "previewTruthIndex",
               "previewFreeOrbit", "previewFreeOrbitSpeed", "previewFreeDistance", "previewFreeFovDeg", "previewFreeRotX",
               "previewFreeRotY", "renderResX", "renderResY"]   # src/Project.h:64-73, in order


_FLOAT_MEMBERS = {f.name for f in dataclasses.fields(Project) if f.type in (float, "float")}


def _settings_text(project):
    """What `file << j` writes (src/ui/UiFrame.cpp:326-330): ONE compact line, keys sorted (nlohmann::json's object is a std::map), every
    float the shortest digits of double(float(member)) — the member is a C++ float.  include/gsplat_extras.hpp writes the same bytes."""
    f32 = lambda v: float(np.float32(v))
    j = {}
    for k in _SERIALISED:
        v = getattr(project, k)
        if isinstance(v, CameraSphere):
            j[k] = dict(count=int(v.count), distance=f32(v.distance), fovDeg=f32(v.fovDeg), rotX=f32(v.rotX), rotY=f32(v.rotY))
        elif isinstance(v, bool) or isinstance(v, str):
            j[k] = v
        elif k in _FLOAT_MEMBERS:
            j[k] = f32(v)
        else:
            j[k] = int(v)
    return json.dumps(j, sort_keys=True, separators=(",", ":"), ensure_ascii=False)


def saveSettings(path, project):
    """UiFrame::saveSettings, src/ui/UiFrame.cpp:323-331."""
    with open(path, "w", encoding="utf-8") as f:
        f.write(_settings_text(project))


def loadSettings(path, project=None):
    """UiFrame::loadSettings, src/ui/UiFrame.cpp:360-371: from_json WITH_DEFAULT (src/Project.h:22,64) — a key absent from the file gets
    the value of a DEFAULT-CONSTRUCTED Project / CameraSphere (not the loaded-into object's current one), unknown keys are ignored, a value
    of the wrong JSON type raises.  `project` (optional) is filled in place and returned."""
    with open(path, encoding="utf-8") as f:
        j = json.load(f)
    if not isinstance(j, dict):
        raise RuntimeError("settings.json: the top level must be an object")
    fresh = Project()

    def number(key, v, as_int):
        if isinstance(v, bool) or not isinstance(v, (int, float)):
            raise RuntimeError(f'settings.json: "{key}" must be a number')
        return int(v) if as_int else float(np.float32(v))
    for k in _SERIALISED:
        default = getattr(fresh, k)
        if k not in j:
            value = default
        elif isinstance(default, CameraSphere):
            if not isinstance(j[k], dict):
                raise RuntimeError(f'settings.json: "{k}" must be an object')
            value = CameraSphere()
            for kk in ("count", "distance", "fovDeg", "rotX", "rotY"):
                if kk in j[k]:
                    setattr(value, kk, number(f"{k}.{kk}", j[k][kk], kk == "count"))
        elif isinstance(default, bool):
            if not isinstance(j[k], bool):
                raise RuntimeError(f'settings.json: "{k}" must be true or false')
            value = j[k]
        elif isinstance(default, str):
            if not isinstance(j[k], str):
                raise RuntimeError(f'settings.json: "{k}" must be a string')
            value = j[k]
        else:
            value = number(k, j[k], k not in _FLOAT_MEMBERS)
        if project is not None:
            setattr(project, k, value)
        else:
            setattr(fresh, k, value)
    return project if project is not None else fresh


def saveCheckpoint(path, model, moment1=None, moment2=None, adam_steps=0, project=None):
    """A lossless checkpoint of a run — what the reference's two files cannot hold: `.gobj` rounds every float to 6 significant
    digits (src/ui/UiFrame.cpp:333-358) and knows no optimizer state (its update rule has none).  One `.npz`:
    the five fp32 arrays of `model` (a ModelSplatsHost) bit for bit with its capacity / SH fields, the two Adam moments and the
    step counter as Trainer.adam_state() returns them (optional), and the Project as its settings.json text (optional).
    numpy's own loader reads it back without executing anything (allow_pickle stays False)."""
    M, n = model.shCoeffs, model.count
    arrays = dict(locations=np.asarray(model.locations[:3 * n], np.float32), shs=np.asarray(model.shs[:3 * M * n], np.float32),
                  scales=np.asarray(model.scales[:3 * n], np.float32), opacities=np.asarray(model.opacities[:n], np.float32),
                  rotations=np.asarray(model.rotations[:4 * n], np.float32),
                  meta=np.array([model.capacity, model.shDegree, M, n, int(adam_steps)], np.int64))
    if moment1 is not None:
        arrays["adam_moment1"] = np.asarray(moment1, np.float32)
        arrays["adam_moment2"] = np.asarray(moment2, np.float32)
    if project is not None:
        d = dataclasses.asdict(project)
        for k in ("updateRule", "adamBeta1", "adamBeta2", "adamEps", "quatLayout"):   # build-side fields travel too, under their own key
            arrays.setdefault("extras_" + k, np.array([d.pop(k)], np.float64))
        arrays["settings_json"] = np.frombuffer(json.dumps(d).encode(), np.uint8)
    np.savez(path, **arrays)


def loadCheckpoint(path, project=None):
    """-> (ModelSplatsHost, moment1 | None, moment2 | None, adam_steps, Project | None); the counterpart of saveCheckpoint.
    Resume: `trainer.model = ModelSplatsDevice(host); trainer.set_adam_state(m1, m2, steps)` continues the run bit for bit
    (tests/test_gpu_trainer.py::test_adam_state_restore_resumes_bit_exact)."""
    z = np.load(path, allow_pickle=False)
    cap, deg, M, n, steps = (int(x) for x in z["meta"])
    host = ModelSplatsHost(cap, deg, M)
    host.locations[:3 * n] = z["locations"]; host.shs[:3 * M * n] = z["shs"]; host.scales[:3 * n] = z["scales"]
    host.opacities[:n] = z["opacities"]; host.rotations[:4 * n] = z["rotations"]
    host.count = n
    m1 = z["adam_moment1"] if "adam_moment1" in z.files else None
    m2 = z["adam_moment2"] if "adam_moment2" in z.files else None
    proj = None
    if "settings_json" in z.files:
        d = json.loads(bytes(z["settings_json"]).decode())
        proj = project or Project()
        for k, v in d.items():
            if k in ("sphere1", "sphere2"):
                setattr(proj, k, CameraSphere(**v))
            elif hasattr(proj, k):
                setattr(proj, k, v)
        for k in ("updateRule", "quatLayout"):
            if "extras_" + k in z.files:
                setattr(proj, k, int(z["extras_" + k][0]))
        for k in ("adamBeta1", "adamBeta2", "adamEps"):
            if "extras_" + k in z.files:
                setattr(proj, k, float(z["extras_" + k][0]))
    return host, m1, m2, steps, proj
