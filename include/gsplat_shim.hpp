// gsplat_shim.hpp — header-only C++ re-creation of the reference's three hot-path classes on top of
// the C-ABI (gsplat.h), so that the reference's UI code keeps compiling against the same names:
//
//     ModelSplatsHost    src/ModelSplatsHost.h:8-39,  src/ModelSplatsHost.cpp:6-91
//     ModelSplatsDevice  src/ModelSplatsDevice.h:5-30, src/ModelSplatsDevice.cpp:6-48
//     Trainer            src/Trainer.cuh:10-75,       src/Trainer.cu:103-543
//
// Same public members, same argument meaning, same error behaviour (std::runtime_error with the
// reference's messages).  Differences a maintainer has to know (INTEGRATION.md has the diff):
//   * glm types in signatures are replaced by plain float arrays (this header has no dependencies);
//   * ModelSplatsDevice owns an opaque gs_model* instead of five raw device pointers — device data
//     is SoA inside the library; host code reaches it through ModelSplatsHost(const ModelSplatsDevice&);
//   * Trainer::captureTruths takes the truth images as input (the OptiX renderer is out of scope);
//   * Trainer::train takes the hyper-parameters as a gs_hyper (Project's fields, src/Project.h:26-41)
//     and returns the statistics; the caller increments Project::iterations (src/Trainer.cu:255).
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
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

class Trainer {
public:
    // `model` mirrors the reference's public pointer: callers `delete trainer->model; trainer->model = new
    // ModelSplatsDevice(host);` (src/ui/UiFrame.cpp:157-158).  Call adoptModel() after assigning.
    ModelSplatsDevice* model = nullptr;
    std::vector<std::vector<uint32_t>> truthFrameBuffersW;  // host copies (the reference keeps device pointers)
    std::vector<std::vector<uint32_t>> truthFrameBuffersB;
    std::vector<gs_view> truthViewsW, truthViewsB;          // per camera: the pass parameters (white / black)

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

    // After `trainer->model = new ModelSplatsDevice(host)`: hand the device model to the library.
    void adoptModel() {
        check(gs_trainer_set_model(handle, model->release()));
        viewsDirty = true;
    }

    // Trainer::render, src/Trainer.cu:148-216.  `view` carries the camera (view/projview/campos/tan_fov*).
    void render(uint32_t* frameBuffer, int sizeX, int sizeY, float splatScale, const gs_view& view, bool frameBufferOnDevice = false) {
        check(gs_trainer_render(handle, frameBuffer, frameBufferOnDevice ? 1 : 0, sizeX, sizeY, splatScale, &view));
    }

    // Replaces Trainer::captureTruths (src/Trainer.cu:218-250): per camera, the white- and black-background
    // pass parameters and RGBA8 truth images (width*height each).
    void captureTruths(const std::vector<gs_view>& viewsWhite, const std::vector<gs_view>& viewsBlack,
                       const std::vector<std::vector<uint32_t>>& framesWhite, const std::vector<std::vector<uint32_t>>& framesBlack) {
        truthViewsW = viewsWhite; truthViewsB = viewsBlack;
        truthFrameBuffersW = framesWhite; truthFrameBuffersB = framesBlack;
        viewsDirty = true;
    }

    // Trainer::train(Project&, bool densify), src/Trainer.cu:252-543.
    gs_step_stats train(const gs_hyper& hyper, bool densify) {
        if (truthFrameBuffersW.empty()) throw std::runtime_error("Can't run training iteration, no truth data available!");
        if (viewsDirty) uploadViews();
        gs_step_stats st{};
        check(gs_trainer_step(handle, &hyper, densify ? 1 : 0, &st));
        if (densify) model->refresh();
        return st;
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
