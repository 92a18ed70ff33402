// gsplat_shim.hpp — header-only C++ re-creation of the reference's three hot-path classes on top of
// the C-ABI (gsplat.h), so that the reference's UI code keeps compiling against the same names:
//
//     ModelSplatsHost    src/ModelSplatsHost.h:8-39,  src/ModelSplatsHost.cpp:6-91
//     ModelSplatsDevice  src/ModelSplatsDevice.h:5-30, src/ModelSplatsDevice.cpp:6-48
//     Trainer            src/Trainer.cuh:10-75,       src/Trainer.cu:103-543
//
// plus the two small value types the Trainer's signatures use:
//     Camera             src/Camera.h:9-26,  getView / getProjection src/Camera.cpp:79-86
//     Project            src/Project.h:6-75  (run-time hyper-parameters; any type with the same field names works)
//
// Same public members, same argument meaning, same error behaviour (std::runtime_error with the
// reference's messages).  The reference's call sites compile unchanged:
//     delete trainer->model; trainer->model = new ModelSplatsDevice(host);      (src/ui/UiFrame.cpp:157-158)
//     trainer->train(project, densify);                                          (src/ui/UiFrame.cpp:288)
//     trainer->render(fb, w, h, project.previewSplatScale, camera);              (src/ui/UiPanelViewOutput.cpp:52-60)
//     trainer->truthCameras.size()                                               (src/ui/UiPanelViewInput.cpp:39)
// Differences a maintainer has to know (INTEGRATION.md has the diff):
//   * glm types in signatures are replaced by plain float arrays (this header has no dependencies);
//   * ModelSplatsDevice owns an opaque gs_model* instead of five raw device pointers — device data
//     is SoA inside the library; host code reaches it through ModelSplatsHost(const ModelSplatsDevice&);
//   * Trainer::captureTruths takes the truth images as input (the OptiX renderer is out of scope);
//   * truthFrameBuffersW/B hold host copies of the truth images (the reference keeps device pointers).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "gsplat.h"

namespace gsplat_shim {

inline void check(int status) {
    if (status != GS_OK) {
        const char* msg = gs_last_error();
        throw std::runtime_error((msg && *msg) ? std::string(msg) : std::string(gs_status_string(status)));
    }
}

class ModelSplatsDevice;

class ModelSplatsHost {
public:
    int capacity;
    int shDegree;
    int shCoeffs;
    int count = 0;
    float* locations = nullptr;
    float* shs = nullptr;
    float* scales = nullptr;
    float* opacities = nullptr;
    float* rotations = nullptr;

    ModelSplatsHost(int capacity_, int shDegree_, int shCoeffs_) : capacity(capacity_), shDegree(shDegree_), shCoeffs(shCoeffs_) {
        allocate();
    }
    explicit ModelSplatsHost(const ModelSplatsDevice& device);  // defined below
    ModelSplatsHost(const std::vector<float>& locationsArg, const std::vector<float>& shsArg, const std::vector<float>& scalesArg,
                    const std::vector<float>& opacitiesArg, const std::vector<float>& rotationsArg) {
        capacity = 1000000;
        while ((size_t)capacity < locationsArg.size() / 3) capacity *= 10;
        count = (int)locationsArg.size() / 3;
        if (count == 0) throw std::runtime_error("Inconsistent feature dimensions supplied when creating a host model!");
        shDegree = (((int)shsArg.size() / (3 * count)) - 1) / 3;  // the reference's formula, src/ModelSplatsHost.cpp:36
        shCoeffs = ((int)shsArg.size() / (3 * count));
        if (locationsArg.size() != (size_t)count * 3 || shsArg.size() != (size_t)count * 3 * shCoeffs ||
            scalesArg.size() != (size_t)count * 3 || opacitiesArg.size() != (size_t)count || rotationsArg.size() != (size_t)count * 4) {
            throw std::runtime_error("Inconsistent feature dimensions supplied when creating a host model!");
        }
        allocate();
        std::memcpy(locations, locationsArg.data(), (size_t)count * 3 * sizeof(float));
        std::memcpy(shs, shsArg.data(), (size_t)count * 3 * shCoeffs * sizeof(float));
        std::memcpy(scales, scalesArg.data(), (size_t)count * 3 * sizeof(float));
        std::memcpy(opacities, opacitiesArg.data(), (size_t)count * sizeof(float));
        std::memcpy(rotations, rotationsArg.data(), (size_t)count * 4 * sizeof(float));
    }
    ModelSplatsHost(const ModelSplatsHost&) = delete;
    ModelSplatsHost& operator=(const ModelSplatsHost&) = delete;
    ~ModelSplatsHost() {
        delete[] locations; delete[] shs; delete[] scales; delete[] opacities; delete[] rotations;
    }

    // rotation: the 4 floats exactly as the reference memcpy's its glm::quat (src/ModelSplatsHost.cpp:74)
    void pushBack(const float location[3], const std::vector<float>& sh, const float scale[3], float opacity, const float rotation[4]) {
        if (count >= capacity) throw std::runtime_error("Model ran out of capacity!");
        std::memcpy(&locations[count * 3], location, 3 * sizeof(float));
        for (int i = 0; i < shCoeffs * 3; i++) shs[(size_t)count * 3 * shCoeffs + i] = sh.at(i);
        std::memcpy(&scales[count * 3], scale, 3 * sizeof(float));
        opacities[count] = opacity;
        std::memcpy(&rotations[count * 4], rotation, 4 * sizeof(float));
        count++;
    }
    void copy(int indexTo, int indexFrom) {
        if (indexTo < 0 || indexTo >= count || indexFrom < 0 || indexFrom >= count)
            throw std::runtime_error("Can't copy splat in model, incorrect bounds and/or no capacity!");
        std::memcpy(&locations[indexTo * 3], &locations[indexFrom * 3], 3 * sizeof(float));
        for (int i = 0; i < shCoeffs * 3; i++) shs[(size_t)indexTo * 3 * shCoeffs + i] = shs[(size_t)indexFrom * 3 * shCoeffs + i];
        std::memcpy(&scales[indexTo * 3], &scales[indexFrom * 3], 3 * sizeof(float));
        opacities[indexTo] = opacities[indexFrom];
        std::memcpy(&rotations[indexTo * 4], &rotations[indexFrom * 4], 4 * sizeof(float));
    }

private:
    void allocate() {
        locations = new float[(size_t)capacity * 3];
        shs = new float[(size_t)capacity * 3 * shCoeffs];
        scales = new float[(size_t)capacity * 3];
        opacities = new float[(size_t)capacity];
        rotations = new float[(size_t)capacity * 4];
    }
};

class ModelSplatsDevice {
public:
    int capacity;
    int shDegree;
    int shCoeffs;
    int count = 0;
    gs_model* handle = nullptr;  // replaces devLocations / devShs / devScales / devOpacities / devRotations

    ModelSplatsDevice(const ModelSplatsDevice& device) : capacity(device.capacity), shDegree(device.shDegree), shCoeffs(device.shCoeffs) {
        count = device.count;
        check(gs_model_clone(device.handle, &handle));
    }
    explicit ModelSplatsDevice(const ModelSplatsHost& host) : capacity(host.capacity), shDegree(host.shDegree), shCoeffs(host.shCoeffs) {
        count = host.count;
        check(gs_model_create(capacity, shDegree, shCoeffs, count, host.locations, host.shs, host.scales, host.opacities, host.rotations, &handle));
    }
    // adopt a handle the trainer owns (used by Trainer::model after densify changed the count)
    ModelSplatsDevice(gs_model* borrowed, bool owns) : handle(borrowed), owned(owns) { refresh(); }
    ModelSplatsDevice& operator=(const ModelSplatsDevice&) = delete;
    ~ModelSplatsDevice() { if (owned && handle) gs_model_destroy(handle); }

    void refresh() { check(gs_model_info(handle, &capacity, &shDegree, &shCoeffs, &count)); }
    gs_model* release() { owned = false; return handle; }

private:
    bool owned = true;
};

inline ModelSplatsHost::ModelSplatsHost(const ModelSplatsDevice& device) : ModelSplatsHost(device.capacity, device.shDegree, device.shCoeffs) {
    count = device.count;
    check(gs_model_download(device.handle, locations, shs, scales, opacities, rotations));
}

// src/Camera.h:9-26.  Matrices come out as glm column-major float[16] (m[col * 4 + row]).
class Camera {
public:
    float location[3];
    float target[3];
    float fovDegY;

    Camera(const float locationArg[3], const float targetArg[3], float fovDegYArg) : fovDegY(fovDegYArg) {
        for (int k = 0; k < 3; k++) { location[k] = locationArg[k]; target[k] = targetArg[k]; }
    }
    Camera(float lx, float ly, float lz, float tx, float ty, float tz, float fovDegYArg) : location{ lx, ly, lz }, target{ tx, ty, tz }, fovDegY(fovDegYArg) {}

    // -glm::lookAt(location, target, +Y), src/Camera.cpp:79-82 (every entry negated: view-space +z is forward)
    void getView(float out[16]) const {
        const float up[3] = { 0.0f, 1.0f, 0.0f };
        float f[3], s[3], u[3];
        for (int k = 0; k < 3; k++) f[k] = target[k] - location[k];
        unit(f);
        cross(f, up, s);
        unit(s);
        cross(s, f, u);
        const float m[16] = { s[0], u[0], -f[0], 0.0f, s[1], u[1], -f[1], 0.0f, s[2], u[2], -f[2], 0.0f,
                              -dot(s, location), -dot(u, location), dot(f, location), 1.0f };
        for (int k = 0; k < 16; k++) out[k] = -m[k];
    }
    // glm::perspective(radians(fovDegY), aspect, 0.1, 100), src/Camera.cpp:84-86
    void getProjection(float aspect, float out[16]) const {
        const float fovy = fovDegY * 0.01745329251994329576923690768489f, zNear = 0.1f, zFar = 100.0f;
        const float t = std::tan(fovy / 2.0f);
        for (int k = 0; k < 16; k++) out[k] = 0.0f;
        out[0] = 1.0f / (aspect * t);
        out[5] = 1.0f / t;
        out[10] = -(zFar + zNear) / (zFar - zNear);
        out[11] = -1.0f;
        out[14] = -(2.0f * zFar * zNear) / (zFar - zNear);
    }
    // Camera::getCamerasCount / getCameras / getPreviewCamera, src/Camera.cpp:29-74.  Templates over the project type: the
    // reference's own Project (src/Project.h) fits, and so does gsplat_shim::Project below.
    template <class ProjectT> static int getCamerasCount(const ProjectT& project) { return project.sphere1.count + project.sphere2.count; }
    // two Fibonacci spheres (:9-27), each turned by angleAxis(radians(rotX), +Y) * angleAxis(radians(rotY), +X) (:40-41,49-50)
    template <class ProjectT> static std::vector<Camera> getCameras(const ProjectT& project) {
        std::vector<Camera> out;
        out.reserve((size_t)getCamerasCount(project));
        const auto sphere = [&out](int count, float distance, float fovDeg, float rotX, float rotY) {
            float rot[3][3];
            orbit(rotX * kRadians, rotY * kRadians, rot);
            const float goldenRatio = (1.0f + std::sqrt(5.0f)) / 2.0f;
            const float angleStep = 2.0f * 3.14159265358979323846f * goldenRatio;
            for (int i = 0; i < count; i++) {
                const float t = (float)i / (float)count;
                const float angle1 = std::acos(1.0f - 2.0f * t), angle2 = angleStep * (float)i;
                const float p[3] = { std::sin(angle1) * std::cos(angle2) * distance, std::sin(angle1) * std::sin(angle2) * distance, std::cos(angle1) * distance };
                out.push_back(turned(rot, p, fovDeg));
            }
        };
        sphere(project.sphere1.count, project.sphere1.distance, project.sphere1.fovDeg, project.sphere1.rotX, project.sphere1.rotY);
        sphere(project.sphere2.count, project.sphere2.distance, project.sphere2.fovDeg, project.sphere2.rotX, project.sphere2.rotY);
        return out;
    }
    // The camera every Trainer::render caller of the reference builds (src/ui/UiPanelViewOutput.cpp:52-60,
    // src/ui/tools/UiPanelToolsView.cpp:250): one of the truth cameras (previewTruth, .at() throws past the end like the
    // reference's), or the free camera — (0, 0, -previewFreeDistance) turned by angleAxis(radians(previewFreeRotY) + orbit, +Y)
    // * angleAxis(radians(previewFreeRotX), +X), where the reference adds previewTimer * previewFreeOrbitSpeed to the angle
    // AFTER the conversion to radians (src/Camera.cpp:68-69: the "degRotOrbit" term is used as radians; kept as is).
    template <class ProjectT> static Camera getPreviewCamera(const ProjectT& project) {
        if (project.previewTruth) return getCameras(project).at((size_t)project.previewTruthIndex);
        const float degRotOrbit = project.previewFreeOrbit ? project.previewTimer * project.previewFreeOrbitSpeed : 0.0f;
        float rot[3][3];
        orbit(project.previewFreeRotY * kRadians + degRotOrbit, project.previewFreeRotX * kRadians, rot);
        const float p[3] = { 0.0f, 0.0f, -project.previewFreeDistance };
        return turned(rot, p, project.previewFreeFovDeg);
    }

    // One pass of this camera as the trainer passes it to the rasterizer (src/Trainer.cu:317-326, :355-356): view,
    // projection * view, camera position, tan(fov / 2) on both axes, background.
    gs_view pass(int width, int height, float background) const {
        gs_view v{};
        float proj[16];
        getView(v.view);
        getProjection((float)width / (float)height, proj);
        for (int c = 0; c < 4; c++)
            for (int r = 0; r < 4; r++)
                v.projview[c * 4 + r] = proj[0 * 4 + r] * v.view[c * 4 + 0] + proj[1 * 4 + r] * v.view[c * 4 + 1] +
                                        proj[2 * 4 + r] * v.view[c * 4 + 2] + proj[3 * 4 + r] * v.view[c * 4 + 3];
        for (int k = 0; k < 3; k++) { v.campos[k] = location[k]; v.bg[k] = background; }
        v.tan_fovx = v.tan_fovy = std::tan(fovDegY * 0.01745329251994329576923690768489f * 0.5f);
        return v;
    }

private:
    static constexpr float kRadians = 0.01745329251994329576923690768489f;  // glm::radians
    // (glm::mat4)glm::angleAxis(angle, axis) for a unit axis, as a row-major 3x3
    static void angleAxis(float angle, float ax, float ay, float az, float m[3][3]) {
        const float sn = std::sin(angle * 0.5f), w = std::cos(angle * 0.5f), x = ax * sn, y = ay * sn, z = az * sn;
        m[0][0] = 1.0f - 2.0f * (y * y + z * z); m[0][1] = 2.0f * (x * y - w * z); m[0][2] = 2.0f * (x * z + w * y);
        m[1][0] = 2.0f * (x * y + w * z); m[1][1] = 1.0f - 2.0f * (x * x + z * z); m[1][2] = 2.0f * (y * z - w * x);
        m[2][0] = 2.0f * (x * z - w * y); m[2][1] = 2.0f * (y * z + w * x); m[2][2] = 1.0f - 2.0f * (x * x + y * y);
    }
    // angleAxis(aboutY, +Y) * angleAxis(aboutX, +X)
    static void orbit(float aboutY, float aboutX, float out[3][3]) {
        float a[3][3], b[3][3];
        angleAxis(aboutY, 0.0f, 1.0f, 0.0f, a);
        angleAxis(aboutX, 1.0f, 0.0f, 0.0f, b);
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) out[r][c] = a[r][0] * b[0][c] + a[r][1] * b[1][c] + a[r][2] * b[2][c];
    }
    static Camera turned(const float rot[3][3], const float p[3], float fovDeg) {
        return Camera(rot[0][0] * p[0] + rot[0][1] * p[1] + rot[0][2] * p[2], rot[1][0] * p[0] + rot[1][1] * p[1] + rot[1][2] * p[2],
                      rot[2][0] * p[0] + rot[2][1] * p[1] + rot[2][2] * p[2], 0.0f, 0.0f, 0.0f, fovDeg);
    }
    static float dot(const float a[3], const float b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
    static void cross(const float a[3], const float b[3], float o[3]) {
        o[0] = a[1] * b[2] - b[1] * a[2]; o[1] = a[2] * b[0] - b[2] * a[0]; o[2] = a[0] * b[1] - b[0] * a[1];
    }
    static void unit(float v[3]) { const float inv = 1.0f / std::sqrt(dot(v, v)); for (int k = 0; k < 3; k++) v[k] *= inv; }
};

// The run-time hyper-parameters of src/Project.h:24-45 under the reference's names and defaults (the GUI / JSON
// fields of the reference's Project are not needed by this path).  Trainer::train is a template over the project
// type, so the reference's own Project class works unchanged.
struct CameraSphere {  // Project::CameraSphere, src/Project.h:14-22
    int count = 16;
    float distance = 10.0f, fovDeg = 60.0f, rotX = 0.0f, rotY = 0.0f;
};
struct Project {   // src/Project.h:6-75: every serialised member, the reference's names and defaults (gsplat_extras.hpp reads / writes its settings.json)
    std::string perspective;                      // (the GUI's window layout; carried through save / load untouched)
    std::string pathModel, pathTextureDiffuse;
    CameraSphere sphere1, sphere2;  // (UiFrame::initProject empties the second one: count 0, fovDeg 30, src/ui/UiFrame.cpp:129-134)
    int rtSamples = 100;
    float lrLocation = 0.00005f, lrSh = 0.0001f, lrScale = 0.00002f, lrOpacity = 0.0001f, lrRotation = 0.000025f;
    float paramScaleMax = 0.3f;
    float paramCullOpacity = 0.005f, paramCullSize = 0.004f, paramDensifyVariance = 2.0f;
    float paramSplitSize = 0.04f, paramSplitDistance = 1.5f, paramSplitScale = 0.8f, paramCloneDistance = 1.6f;
    int iterations = 0;
    int intervalCapture = 50;
    int intervalDensify = 200;
    // the preview camera's fields (src/Project.h:47-58), read by Camera::getPreviewCamera
    float previewTimer = 0.0f;
    int previewRtSamples = 50;
    float previewSplatScale = 1.0f;
    bool previewTruth = false;
    int previewTruthIndex = 0;
    bool previewFreeOrbit = true;
    float previewFreeOrbitSpeed = 0.5f, previewFreeDistance = 10.0f, previewFreeFovDeg = 60.0f, previewFreeRotX = 25.0f, previewFreeRotY = 0.0f;
    int renderResX = 2048, renderResY = 2048;
};

class Trainer {
public:
    // The reference's public pointer.  Callers replace it directly — `delete trainer->model; trainer->model = new
    // ModelSplatsDevice(host);` (src/ui/UiFrame.cpp:157-158,173-174,261-262,448-449) — and the next train / render /
    // captureTruths call hands the new device model to the library (syncModel).
    ModelSplatsDevice* model = nullptr;
    std::vector<std::vector<uint32_t>> truthFrameBuffersW;  // host copies (the reference keeps device pointers)
    std::vector<std::vector<uint32_t>> truthFrameBuffersB;
    std::vector<Camera> truthCameras;                        // src/Trainer.cuh:54
    std::vector<gs_view> truthViewsW, truthViewsB;          // per camera: the pass parameters (white / black)
    // build-side extensions to the reference's Project: not part of its JSON
    int updateRule = GS_UPDATE_SGD_CLAMP;
    float adamBeta1 = 0.9f, adamBeta2 = 0.999f, adamEps = 1e-15f;
    int quatLayout = GS_QUAT_XYZW;

    Trainer(int width = 1024, int height = 1024) : w(width), h(height) {
        check(gs_trainer_create(width, height, &handle));
        model = new ModelSplatsDevice(gs_trainer_get_model(handle), false);
    }
    Trainer(const Trainer&) = delete;
    Trainer& operator=(const Trainer&) = delete;
    ~Trainer() {
        delete model;
        gs_trainer_destroy(handle);
    }

    // The device model the caller assigned to `model` becomes the library's (ownership moves to the trainer, as in
    // the reference, whose destructor deletes `model`); a no-op while `model` still is the library's own.
    void syncModel() {
        if (!model) throw std::runtime_error("trainer has no model");
        if (model->handle != gs_trainer_get_model(handle)) check(gs_trainer_set_model(handle, model->release()));
    }
    void adoptModel() { syncModel(); }  // (kept for callers written against the first version of this header)

    // Trainer::render, src/Trainer.cu:148-216: black background, tan_fovx = tan(radians(sizeX * fovY / sizeY) / 2) (:196).
    void render(uint32_t* frameBuffer, int sizeX, int sizeY, float splatScale, const Camera& camera, bool frameBufferOnDevice = false) {
        gs_view view = camera.pass(sizeX, sizeY, 0.0f);
        view.tan_fovx = std::tan(((float)sizeX * camera.fovDegY / (float)sizeY) * 0.01745329251994329576923690768489f * 0.5f);
        render(frameBuffer, sizeX, sizeY, splatScale, view, frameBufferOnDevice);
    }
    // the same with the pass parameters spelled out (view / projview / campos / tan_fov* / bg)
    void render(uint32_t* frameBuffer, int sizeX, int sizeY, float splatScale, const gs_view& view, bool frameBufferOnDevice = false) {
        syncModel();
        check(gs_trainer_render(handle, frameBuffer, frameBufferOnDevice ? 1 : 0, sizeX, sizeY, splatScale, &view));
    }

    // Replaces Trainer::captureTruths(const Project&, RtxHost&) (src/Trainer.cu:218-250): the cameras of the scene
    // (Camera::getCameras(project) in the reference) and, per camera, the white- and the black-background RGBA8
    // truth image (width * height each) that the reference's ray tracer would have produced.
    void captureTruths(const std::vector<Camera>& cameras, const std::vector<std::vector<uint32_t>>& framesWhite,
                       const std::vector<std::vector<uint32_t>>& framesBlack) {
        if (cameras.size() != framesWhite.size() || cameras.size() != framesBlack.size()) throw std::runtime_error("captureTruths: one white and one black frame per camera");
        truthCameras = cameras;
        truthViewsW.clear(); truthViewsB.clear();
        for (const Camera& c : cameras) { truthViewsW.push_back(c.pass(w, h, 1.0f)); truthViewsB.push_back(c.pass(w, h, 0.0f)); }
        truthFrameBuffersW = framesWhite; truthFrameBuffersB = framesBlack;
        viewsDirty = true;
    }

    // Replaces Trainer::captureTruths (src/Trainer.cu:218-250): per camera, the white- and black-background
    // pass parameters and RGBA8 truth images (width*height each).
    void captureTruths(const std::vector<gs_view>& viewsWhite, const std::vector<gs_view>& viewsBlack,
                       const std::vector<std::vector<uint32_t>>& framesWhite, const std::vector<std::vector<uint32_t>>& framesBlack) {
        truthViewsW = viewsWhite; truthViewsB = viewsBlack;
        truthFrameBuffersW = framesWhite; truthFrameBuffersB = framesBlack;
        viewsDirty = true;
    }

    // Trainer::train(Project&, bool densify), src/Trainer.cu:252-543: reads the learning rates and densify
    // parameters from the project on every call and counts the iteration (:255).  Like the reference it returns
    // with the device still working unless densify is set.
    // (only types that look like the reference's Project take this overload: a non-const gs_hyper lvalue must still reach the
    // gs_hyper form below)
    template <class ProjectT, class = decltype(std::declval<ProjectT&>().lrLocation), class = decltype(std::declval<ProjectT&>().iterations)>
    void train(ProjectT& project, bool densify) {
        if (truthFrameBuffersW.empty()) throw std::runtime_error("Can't run training iteration, no truth data available!");
        project.iterations++;
        const gs_hyper hyper = hyperOf(project);
        syncModel();
        if (viewsDirty) uploadViews();
        check(gs_trainer_step(handle, &hyper, densify ? 1 : 0, nullptr));
        if (densify) model->refresh();
    }
    // the same with explicit hyper-parameters, returning the step's statistics (this waits for the step to finish)
    gs_step_stats train(const gs_hyper& hyper, bool densify) {
        if (truthFrameBuffersW.empty()) throw std::runtime_error("Can't run training iteration, no truth data available!");
        syncModel();
        if (viewsDirty) uploadViews();
        gs_step_stats st{};
        check(gs_trainer_step(handle, &hyper, densify ? 1 : 0, &st));
        if (densify) model->refresh();
        return st;
    }
    template <class ProjectT> gs_hyper hyperOf(const ProjectT& p) const {
        gs_hyper hy{};
        hy.lr_location = p.lrLocation; hy.lr_sh = p.lrSh; hy.lr_scale = p.lrScale; hy.lr_opacity = p.lrOpacity; hy.lr_rotation = p.lrRotation;
        hy.scale_max = p.paramScaleMax;
        hy.cull_opacity = p.paramCullOpacity; hy.cull_size = p.paramCullSize; hy.densify_variance = p.paramDensifyVariance;
        hy.split_size = p.paramSplitSize; hy.split_distance = p.paramSplitDistance; hy.split_scale = p.paramSplitScale;
        hy.clone_distance = p.paramCloneDistance;
        hy.update_rule = updateRule; hy.adam_beta1 = adamBeta1; hy.adam_beta2 = adamBeta2; hy.adam_eps = adamEps;
        hy.quat_layout = quatLayout;
        return hy;
    }

    // Optimizer state of GS_UPDATE_ADAM for checkpoint / resume (build-side extension: the reference's update rule has no state).
    // adamState: host copies of the two moments + the number of Adam steps (empty vectors before the first Adam step);
    // setAdamState: installs them again — after `model` was assigned, which resets the state.
    int adamState(std::vector<float>& moment1, std::vector<float>& moment2) {
        float *m1 = nullptr, *m2 = nullptr; size_t n = 0; int steps = 0;
        check(gs_trainer_adam_state(handle, &m1, &m2, &n, &steps));
        moment1.assign(m1 ? n : 0, 0.0f); moment2.assign(m2 ? n : 0, 0.0f);
        if (m1) { check(gs_memcpy_d2h(moment1.data(), m1, n * sizeof(float))); check(gs_memcpy_d2h(moment2.data(), m2, n * sizeof(float))); }
        return steps;
    }
    void setAdamState(const std::vector<float>& moment1, const std::vector<float>& moment2, int steps) {
        syncModel();
        if (moment1.size() != moment2.size()) throw std::runtime_error("setAdamState: the two moments differ in size");
        check(gs_trainer_set_adam_state(handle, moment1.empty() ? nullptr : moment1.data(), moment2.empty() ? nullptr : moment2.data(), moment1.size(), steps, 0));
    }

    gs_trainer* native() { return handle; }

private:
    void uploadViews() {
        std::vector<gs_view> views(truthViewsW);
        views.insert(views.end(), truthViewsB.begin(), truthViewsB.end());  // white passes first, then black (src/Trainer.cu:311-314)
        std::vector<const uint32_t*> ptrs;
        for (auto& f : truthFrameBuffersW) ptrs.push_back(f.data());
        for (auto& f : truthFrameBuffersB) ptrs.push_back(f.data());
        check(gs_trainer_set_views(handle, (int)views.size(), views.data(), ptrs.data(), 0, (int)views.size()));
        viewsDirty = false;
    }
    gs_trainer* handle = nullptr;
    int w, h;
    bool viewsDirty = true;
};

}  // namespace gsplat_shim
