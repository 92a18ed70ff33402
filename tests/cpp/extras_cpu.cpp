// extras_cpu.cpp — host-only check of include/gsplat_extras.hpp: grid field, .gobj written by C++ and read back.
#include <cstdio>
#include <cstring>

#include "gsplat_extras.hpp"

using namespace gsplat_shim;

int main(int argc, char** argv) {
    if (argc != 2) return 2;
    auto grid = initFieldGrid();
    if (grid->count != 17 * 17 * 17 || grid->rotations[3] != 1.0f || grid->scales[0] != 0.5f * 0.1f) return 3;
    grid->count = 40;  // keep the file small
    grid->shs[5] = 0.123456789f; grid->opacities[7] = 1e-5f;
    saveSplats(argv[1], *grid);
    auto back = loadSplats(argv[1]);
    if (back->count != 40 || back->shCoeffs != 4 || back->shDegree != 1 || back->capacity != 1000000) return 4;
    if (std::fabs(back->shs[5] - 0.123457f) > 1e-7f || std::fabs(back->opacities[7] - 1e-5f) > 1e-11f) return 5;
    if (initFieldMono()->count != 1) return 6;
    bool threw = false;
    try { ModelSplatsHost bad(std::vector<float>{ 0, 0, 0 }, std::vector<float>{ 1, 2, 3 }, std::vector<float>{ 1, 1 }, { 1 }, { 1, 0, 0, 0 }); }
    catch (const std::runtime_error&) { threw = true; }
    if (!threw) return 7;
    printf("extras ok\n");
    return 0;
}
