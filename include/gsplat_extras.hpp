// gsplat_extras.hpp — header-only C++ versions of the rows either side of the training path:
//   initFieldGrid / initFieldMono   src/ui/UiFrame.cpp:137-176   (canonical start states)
//   saveSplats / loadSplats         src/ui/UiFrame.cpp:333-358, :373-450   (the text .gobj format)
// on top of gsplat_shim.hpp's ModelSplatsHost.  No GUI, no progress dialogs; same file format, same
// error behaviour (std::runtime_error("Inconsistent SH degree!"), the five-vector constructor's checks).
#pragma once
#include <cmath>
#include <fstream>
#include <memory>
#include <optional>
#include <sstream>
#include <string>

#include "gsplat_shim.hpp"

namespace gsplat_shim {

constexpr int SPLATS_LIMIT = 1000000;  // src/Config.h:17
constexpr int SPLATS_SH_DEGREE = 1;    // src/Config.h:19
constexpr int SPLATS_SH_COEF = 4;      // src/Config.h:20

// glm::angleAxis(angle, axis) as the four floats a memcpy of the glm::quat object yields.  glm's default member
// order is {x, y, z, w}; pass quatXYZW = false for GLM_FORCE_QUAT_DATA_WXYZ builds.
inline void angleAxisMemory(float angle, const float axis[3], float out[4], bool quatXYZW = true) {
    const float s = std::sin(angle * 0.5f), w = std::cos(angle * 0.5f);
    const float x = axis[0] * s, y = axis[1] * s, z = axis[2] * s;
    if (quatXYZW) { out[0] = x; out[1] = y; out[2] = z; out[3] = w; }
    else { out[0] = w; out[1] = x; out[2] = y; out[3] = z; }
}

// UiFrame::initFieldGrid, src/ui/UiFrame.cpp:137-160
inline std::unique_ptr<ModelSplatsHost> initFieldGrid(bool quatXYZW = true) {
    auto modelHost = std::make_unique<ModelSplatsHost>(SPLATS_LIMIT, SPLATS_SH_DEGREE, SPLATS_SH_COEF);
    static const float dim = 4.0f;
    static const float step = 0.5f;
    std::vector<float> shs(3 * (size_t)modelHost->shCoeffs, 0.0f);
    const float up[3] = { 0.0f, 1.0f, 0.0f };
    float rot[4];
    angleAxisMemory(0.0f, up, rot, quatXYZW);
    const float scale[3] = { step * 0.1f, step * 0.1f, step * 0.1f };
    for (float x = -dim; x <= dim; x += step)
        for (float y = -dim; y <= dim; y += step)
            for (float z = -dim; z <= dim; z += step) {
                const float loc[3] = { x, y, z };
                modelHost->pushBack(loc, shs, scale, 1.0f, rot);
            }
    return modelHost;
}

// UiFrame::initFieldMono, src/ui/UiFrame.cpp:162-176
inline std::unique_ptr<ModelSplatsHost> initFieldMono(bool quatXYZW = true) {
    auto modelHost = std::make_unique<ModelSplatsHost>(SPLATS_LIMIT, SPLATS_SH_DEGREE, SPLATS_SH_COEF);
    std::vector<float> shs(3 * (size_t)modelHost->shCoeffs, 0.0f);
    const float up[3] = { 0.0f, 1.0f, 0.0f }, loc[3] = { 0.0f, 0.0f, 0.0f }, scale[3] = { 0.3f, 0.3f, 0.3f };
    float rot[4];
    angleAxisMemory(0.0f, up, rot, quatXYZW);
    modelHost->pushBack(loc, shs, scale, 1.0f, rot);
    return modelHost;
}

// UiFrame::saveSplats, src/ui/UiFrame.cpp:333-358 (ostream default formatting: 6 significant digits)
inline void saveSplats(const std::string& path, const ModelSplatsHost& model) {
    std::ofstream file(path);
    for (int i = 0; i < model.count; i++) {
        file << "v " << model.locations[i * 3] << " " << model.locations[i * 3 + 1] << " " << model.locations[i * 3 + 2] << "\n";
        file << "sh";
        for (int f = 0; f < model.shCoeffs * 3; f++) file << " " << model.shs[(size_t)i * 3 * model.shCoeffs + f];
        file << "\n";
        file << "s " << model.scales[i * 3] << " " << model.scales[i * 3 + 1] << " " << model.scales[i * 3 + 2] << "\n";
        file << "a " << model.opacities[i] << "\n";
        file << "r " << model.rotations[i * 4] << " " << model.rotations[i * 4 + 1] << " " << model.rotations[i * 4 + 2] << " "
             << model.rotations[i * 4 + 3] << "\n";
    }
}

// UiFrame::loadSplats, src/ui/UiFrame.cpp:373-450
inline std::unique_ptr<ModelSplatsHost> loadSplats(const std::string& path) {
    std::optional<int> shCoeffs;
    std::vector<float> locations, shs, scales, opacities, rotations;
    std::ifstream file(path);
    if (!file) throw std::runtime_error("Failed to load splats file at \"" + path + "\"!");
    std::string line;
    while (std::getline(file, line)) {
        std::istringstream iss(line);
        std::string prefix;
        iss >> prefix;
        float x;
        if (prefix == "v") { for (int f = 0; f < 3; f++) { iss >> x; locations.push_back(x); } }
        else if (prefix == "sh") {
            int n = 0;
            while (iss >> x) { shs.push_back(x); n++; }
            if (!shCoeffs) shCoeffs = n;
            else if (shCoeffs != n) throw std::runtime_error("Inconsistent SH degree!");
        } else if (prefix == "s") { for (int f = 0; f < 3; f++) { iss >> x; scales.push_back(x); } }
        else if (prefix == "a") { iss >> x; opacities.push_back(x); }
        else if (prefix == "r") { for (int f = 0; f < 4; f++) { iss >> x; rotations.push_back(x); } }
    }
    return std::make_unique<ModelSplatsHost>(locations, shs, scales, opacities, rotations);
}

}  // namespace gsplat_shim
