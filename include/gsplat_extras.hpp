// gsplat_extras.hpp — header-only C++ for the rows either side of the training path, on top
// of gsplat_shim.hpp's ModelSplatsHost:
//   * the text `.gobj` splat format of the reference's save / load menu entries (src/ui/UiFrame.cpp:333-358, :373-450)
//   * the three start fields: grid, mono, one splat per OBJ triangle (src/ui/UiFrame.cpp:137-264)
// No GUI, no progress dialogs.  Same file format, same constants, same error behaviour
// (std::runtime_error("Inconsistent SH degree!"), "Unexpected vertex count in face list!", the five-vector
// constructor's dimension checks).
//
// Structure (this file's own, mirrored by gaussian-splatterer_amd/io.py and fields.py):
//   GobjSchema      the five record kinds of a splat as a table {tag, floats per splat, array}: the writer and the
//                   reader are loops over that table
//   SplatRecord     one splat's five attribute groups, the unit every field initialiser emits
//   lattice / mono / triangles   generators of SplatRecords; makeField() pushes them into a ModelSplatsHost
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <istream>
#include <memory>
#include <ostream>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

#include "gsplat_shim.hpp"

namespace gsplat_shim {

constexpr int SPLATS_LIMIT = 1000000;  // src/Config.h:17
constexpr int SPLATS_SH_DEGREE = 1;    // src/Config.h:19
constexpr int SPLATS_SH_COEF = 4;      // src/Config.h:20

// ---------------------------------------------------------------------------------------------------------
// .gobj
// ---------------------------------------------------------------------------------------------------------
// One record kind: the line tag, how many floats a splat has of it (0 = "3 * shCoeffs", the only variable one) and
// where they live in a host model.
struct GobjField {
    const char* tag;
    int fixedWidth;
    float* ModelSplatsHost::*array;
    int width(int shCoeffs) const { return fixedWidth ? fixedWidth : 3 * shCoeffs; }
};
// File order within a splat: v, sh, s, a, r (src/ui/UiFrame.cpp:343-357).
inline const std::array<GobjField, 5>& gobjSchema() {
    static const std::array<GobjField, 5> schema = { {
        { "v", 3, &ModelSplatsHost::locations },
        { "sh", 0, &ModelSplatsHost::shs },
        { "s", 3, &ModelSplatsHost::scales },
        { "a", 1, &ModelSplatsHost::opacities },
        { "r", 4, &ModelSplatsHost::rotations },
    } };
    return schema;
}

// "<tag> f0 f1 ...\n" at the stream's default float formatting (6 significant digits: the format is lossy)
inline void writeRecord(std::ostream& os, const char* tag, const float* values, int n) {
    os << tag;
    for (int k = 0; k < n; k++) os << ' ' << values[k];
    os << '\n';
}

inline void writeSplats(std::ostream& os, const ModelSplatsHost& model) {
    for (int i = 0; i < model.count; i++)
        for (const GobjField& f : gobjSchema()) {
            const int w = f.width(model.shCoeffs);
            writeRecord(os, f.tag, (model.*(f.array)) + (size_t)i * w, w);
        }
}
inline void saveSplats(const std::string& path, const ModelSplatsHost& model) {
    std::ofstream file(path);
    if (!file) throw std::runtime_error("Failed to open splats file \"" + path + "\" for writing!");
    writeSplats(file, model);
}

// Reader: every line is "<tag> numbers..."; a known tag appends up to its width to that attribute's column (the
// variable-width `sh` record takes every number on the line and all `sh` lines must agree), unknown tags and blank
// lines are skipped, as the reference's prefix chain does.  The columns then go through the five-vector constructor,
// which owns the dimension checks.
// One number the way `iss >> x` (float) reads it: leading blanks skipped, then the longest run of decimal-float characters;
// "nan", "inf" and hexadecimal forms are not numbers to a stream.  Returns false (q unchanged) when there is none.
inline bool streamFloat(const char*& q, float& v) {
    const char* p = q;
    while (*p == ' ' || *p == '\t' || *p == '\r') p++;
    char buf[64];
    size_t n = 0;
    while (n + 1 < sizeof(buf) && *p && std::strchr("+-0123456789.eE", *p)) buf[n++] = *p++;
    buf[n] = 0;
    if (n == 0) return false;
    char* end = nullptr;
    v = std::strtof(buf, &end);
    if (end == buf) return false;
    q += (p - n - q) + (end - buf);
    return true;
}

inline std::unique_ptr<ModelSplatsHost> readSplats(std::istream& is) {
    const auto& schema = gobjSchema();
    std::array<std::vector<float>, 5> column;
    int shPerSplat = -1;
    std::string line;
    while (std::getline(is, line)) {
        const char* p = line.c_str();
        while (*p == ' ' || *p == '\t') p++;
        const char* tagEnd = p;
        while (*tagEnd && *tagEnd != ' ' && *tagEnd != '\t' && *tagEnd != '\r') tagEnd++;
        const std::string tag(p, tagEnd);
        for (size_t k = 0; k < schema.size(); k++) {
            if (tag != schema[k].tag) continue;
            const char* q = tagEnd;
            if (schema[k].fixedWidth) {
                // the reference pushes exactly 3 / 1 / 4 values per v, s / a / r line whatever the line holds
                // (src/ui/UiFrame.cpp:404-434: `float x; iss >> x; push_back(x)`; a failed extraction stores 0)
                bool ok = true;
                for (int f = 0; f < schema[k].fixedWidth; f++) {
                    float v = 0.0f;
                    ok = ok && streamFloat(q, v);
                    column[k].push_back(ok ? v : 0.0f);
                }
            } else {
                int got = 0;
                float v;
                while (streamFloat(q, v)) { column[k].push_back(v); got++; }   // `while (iss >> x)`
                if (shPerSplat < 0) shPerSplat = got;
                else if (shPerSplat != got) throw std::runtime_error("Inconsistent SH degree!");
            }
        }
    }
    return std::make_unique<ModelSplatsHost>(column[0], column[1], column[2], column[3], column[4]);
}
inline std::unique_ptr<ModelSplatsHost> loadSplats(const std::string& path) {
    std::ifstream file(path);
    if (!file) throw std::runtime_error("Failed to load splats file at \"" + path + "\"!");
    return readSplats(file);
}

// ---------------------------------------------------------------------------------------------------------
// start fields
// ---------------------------------------------------------------------------------------------------------
struct SplatRecord {
    float location[3];
    float scale[3];
    float opacity;
    float rotation[4];  // the four floats a memcpy of the reference's glm::quat yields (layout: see quatMemory)
};

// glm::angleAxis(angle, axis) as stored by the reference (ModelSplatsHost::pushBack memcpy's the glm::quat object,
// src/ModelSplatsHost.cpp:74).  glm's default member order is {x, y, z, w}; xyzw = false models
// GLM_FORCE_QUAT_DATA_WXYZ builds.  The axis is used as given (glm does not normalise it).
inline void quatMemory(float angle, const float axis[3], bool xyzw, float out[4]) {
    const float half = angle * 0.5f, s = std::sin(half), w = std::cos(half);
    const float v[3] = { axis[0] * s, axis[1] * s, axis[2] * s };
    if (xyzw) { out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; out[3] = w; }
    else { out[0] = w; out[1] = v[0]; out[2] = v[1]; out[3] = v[2]; }
}

using SplatSink = std::function<void(const SplatRecord&)>;
using SplatSource = std::function<void(const SplatSink&)>;

// A fresh model of the compile-time limits (src/Config.h:17-20) filled by a generator; SH all zero (grey 0.5).
inline std::unique_ptr<ModelSplatsHost> makeField(const SplatSource& source) {
    auto model = std::make_unique<ModelSplatsHost>(SPLATS_LIMIT, SPLATS_SH_DEGREE, SPLATS_SH_COEF);
    const std::vector<float> grey(3 * (size_t)model->shCoeffs, 0.0f);
    source([&](const SplatRecord& r) { model->pushBack(r.location, grey, r.scale, r.opacity, r.rotation); });
    return model;
}

// src/ui/UiFrame.cpp:137-160 — a cubic lattice over [-4, 4]^3 with pitch 0.5 (17^3 splats, z fastest), isotropic
// scale pitch / 10, opaque, identity rotation.  Pitch and extent are exact binary fractions, so indexing the lattice
// gives the same coordinates as the reference's accumulating float loops.
inline std::unique_ptr<ModelSplatsHost> initFieldGrid(bool quatXYZW = true) {
    constexpr float extent = 4.0f, pitch = 0.5f;
    constexpr int perAxis = (int)(2.0f * extent / pitch) + 1;
    return makeField([&](const SplatSink& emit) {
        SplatRecord r{};
        const float up[3] = { 0.0f, 1.0f, 0.0f };
        quatMemory(0.0f, up, quatXYZW, r.rotation);
        r.opacity = 1.0f;
        r.scale[0] = r.scale[1] = r.scale[2] = pitch * 0.1f;
        for (int index = 0; index < perAxis * perAxis * perAxis; index++) {
            const int cell[3] = { index / (perAxis * perAxis), (index / perAxis) % perAxis, index % perAxis };
            for (int a = 0; a < 3; a++) r.location[a] = -extent + pitch * (float)cell[a];
            emit(r);
        }
    });
}

// src/ui/UiFrame.cpp:162-176 — one large splat at the origin
inline std::unique_ptr<ModelSplatsHost> initFieldMono(bool quatXYZW = true) {
    return makeField([&](const SplatSink& emit) {
        SplatRecord r{};
        const float up[3] = { 0.0f, 1.0f, 0.0f };
        quatMemory(0.0f, up, quatXYZW, r.rotation);
        r.opacity = 1.0f;
        r.scale[0] = r.scale[1] = r.scale[2] = 0.3f;
        emit(r);
    });
}

// The reference's OBJ subset (src/ui/UiFrame.cpp:185-232): `v x y z`; `f` with 3 or 4 corners `vi[/...]` (1-based;
// quads are fanned from their first corner); every other line ignored; any other corner count is an error.
struct ObjMesh {
    std::vector<std::array<float, 3>> vertices;
    std::vector<std::array<int, 3>> triangles;
};
inline ObjMesh parseObj(std::istream& is) {
    ObjMesh mesh;
    std::string line;
    while (std::getline(is, line)) {
        std::istringstream words(line);
        std::string kind;
        if (!(words >> kind)) continue;
        if (kind == "v") {
            std::array<float, 3> v{};
            words >> v[0] >> v[1] >> v[2];
            mesh.vertices.push_back(v);
        } else if (kind == "f") {
            std::vector<int> corner;
            for (std::string word; words >> word;) corner.push_back(std::atoi(word.substr(0, word.find('/')).c_str()) - 1);
            if (corner.size() != 3 && corner.size() != 4)
                throw std::runtime_error("Unexpected vertex count in face list!" + std::to_string(corner.size()));
            for (size_t k = 2; k < corner.size(); k++) mesh.triangles.push_back({ corner[0], corner[k - 1], corner[k] });
        }
    }
    return mesh;
}

// src/ui/UiFrame.cpp:234-259 — a triangle becomes a thin splat at its centroid: the two edges leaving corner 0 give
// the in-plane extents, 0.005 the thickness (all x 0.2), and +Z is turned onto the face normal by
// angleAxis(acos(n.z), +Z x n).
inline SplatRecord triangleSplat(const std::array<float, 3>& a, const std::array<float, 3>& b, const std::array<float, 3>& c, bool quatXYZW) {
    SplatRecord r{};
    float e1[3], e2[3];
    for (int k = 0; k < 3; k++) {
        r.location[k] = (a[k] + b[k] + c[k]) / 3.0f;
        e1[k] = b[k] - a[k];
        e2[k] = c[k] - a[k];
    }
    auto length = [](const float v[3]) { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); };
    r.scale[0] = length(e1) * 0.2f; r.scale[1] = length(e2) * 0.2f; r.scale[2] = 0.005f * 0.2f;
    float n[3] = { e1[1] * e2[2] - e2[1] * e1[2], e1[2] * e2[0] - e2[2] * e1[0], e1[0] * e2[1] - e2[0] * e1[1] };
    const float inv = 1.0f / length(n);
    for (float& x : n) x *= inv;
    const float axis[3] = { -n[1], n[0], 0.0f };  // (0, 0, 1) x n
    quatMemory(std::acos(n[2]), axis, quatXYZW, r.rotation);
    r.opacity = 1.0f;
    return r;
}

// src/ui/UiFrame.cpp:178-264 — one splat per triangle of an OBJ mesh
inline std::unique_ptr<ModelSplatsHost> initFieldModel(std::istream& obj, bool quatXYZW = true) {
    const ObjMesh mesh = parseObj(obj);
    return makeField([&](const SplatSink& emit) {
        for (const auto& t : mesh.triangles) emit(triangleSplat(mesh.vertices.at(t[0]), mesh.vertices.at(t[1]), mesh.vertices.at(t[2]), quatXYZW));
    });
}
inline std::unique_ptr<ModelSplatsHost> initFieldModel(const std::string& objPath, bool quatXYZW = true) {
    std::ifstream file(objPath);
    if (!file) throw std::runtime_error("Failed to load model file at \"" + objPath + "\"!");
    return initFieldModel(file, quatXYZW);
}

// ---------------------------------------------------------------------------------------------------------------------
// The caller of the path: the "Auto Train" loop of UiFrame::update (src/ui/UiFrame.cpp:266-298) without the GUI — at most
// AUTO_TRAIN_BUDGET = 100 iterations per second (src/Config.h:10; the budget never accumulates beyond one iteration), truth
// re-capture with randomly re-rotated camera spheres every `intervalCapture` iterations (UiPanelToolsTruth::onButtonRandomRotate
// / onButtonCapture, src/ui/tools/UiPanelToolsTruth.cpp:186-197), densify every `intervalDensify` iterations.
// TrainerT needs train(ProjectT&, bool) and captureTruths(cameras, framesWhite, framesBlack) (gsplat_shim::Trainer has both);
// the truth renderer stays the caller's (the reference's is OptiX): capture(cameras, framesWhite, framesBlack) fills one RGBA8
// image per camera and background.  random01: a uniform [0, 1) source (the reference calls rand() / RAND_MAX).
template <class TrainerT, class ProjectT>
class AutoTrainer {
public:
    using Frames = std::vector<std::vector<uint32_t>>;
    using Capture = std::function<void(const std::vector<Camera>&, Frames& white, Frames& black)>;
    static constexpr float kBudgetPerSecond = 100.0f;  // AUTO_TRAIN_BUDGET
    bool autoTraining = true;
    float autoTrainingBudget = 0.0f;

    AutoTrainer(TrainerT& trainerArg, ProjectT& projectArg, Capture captureArg, std::function<float()> random01 = {})
        : trainer(trainerArg), project(projectArg), capture(std::move(captureArg)),
          random(random01 ? std::move(random01) : std::function<float()>([] { return (float)std::rand() / ((float)RAND_MAX + 1.0f); })) {}

    void randomRotate() {  // src/ui/tools/UiPanelToolsTruth.cpp:192-197
        project.sphere1.rotX = random() * 360.0f; project.sphere1.rotY = random() * 360.0f;
        project.sphere2.rotX = random() * 360.0f; project.sphere2.rotY = random() * 360.0f;
    }
    void captureTruths() {  // :186-190 -> Trainer::captureTruths with the project's cameras
        const std::vector<Camera> cameras = Camera::getCameras(project);
        Frames white, black;
        capture(cameras, white, black);
        trainer.captureTruths(cameras, white, black);
    }
    // one iteration: the body of `if(autoTrainingBudget >= 1.0f)` (src/ui/UiFrame.cpp:279-296); returns {captured, densified}
    std::pair<bool, bool> step() {
        const bool cap = project.intervalCapture > 0 && project.iterations % project.intervalCapture == 0;
        const bool densify = project.intervalDensify > 0 && project.iterations % project.intervalDensify == 0;
        if (cap) { randomRotate(); captureTruths(); }
        trainer.train(project, densify);
        return { cap, densify };
    }
    // UiFrame::update (:266-298), called from the host's idle loop with the seconds since the last call; true when an iteration ran
    bool update(float delta) {
        project.previewTimer += delta;
        if (!autoTraining) return false;
        autoTrainingBudget = std::min(1.0f, autoTrainingBudget + delta * kBudgetPerSecond);
        if (autoTrainingBudget < 1.0f) return false;
        autoTrainingBudget = 0.0f;
        step();
        return true;
    }

private:
    TrainerT& trainer;
    ProjectT& project;
    Capture capture;
    std::function<float()> random;
};

}  // namespace gsplat_shim
