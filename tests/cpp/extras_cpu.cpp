// extras_cpu.cpp — host-only check of include/gsplat_extras.hpp.
//   extras_cpu <dir>: reads <dir>/py.gobj (written by the Python mirror) and <dir>/mesh.obj, writes
//     <dir>/cpp_copy.gobj   = load(py.gobj) saved again            (must equal Python's own load -> save, byte for byte)
//     <dir>/cpp_grid.gobj   = the first 40 splats of initFieldGrid
//     <dir>/cpp_mesh.gobj   = initFieldModel(mesh.obj)
#include <cstdio>
#include <cstring>

#include "gsplat_extras.hpp"

using namespace gsplat_shim;

int main(int argc, char** argv) {
    if (argc != 2) return 2;
    const std::string dir = argv[1];
    auto grid = initFieldGrid();
    if (grid->count != 17 * 17 * 17 || grid->rotations[3] != 1.0f || grid->scales[0] != 0.5f * 0.1f) return 3;
    if (grid->locations[0] != -4.0f || grid->locations[5] != -3.5f || grid->locations[3 * (17 * 17 * 17 - 1)] != 4.0f) return 3;  // z fastest
    if (initFieldGrid(false)->rotations[0] != 1.0f) return 3;
    grid->count = 40;  // keep the file small
    saveSplats(dir + "/cpp_grid.gobj", *grid);
    if (initFieldMono()->count != 1 || initFieldMono()->scales[1] != 0.3f) return 6;

    auto copy = loadSplats(dir + "/py.gobj");
    if (copy->shCoeffs != 4 || copy->shDegree != 1 || copy->capacity != 1000000) return 4;
    saveSplats(dir + "/cpp_copy.gobj", *copy);

    auto mesh = initFieldModel(dir + "/mesh.obj");
    saveSplats(dir + "/cpp_mesh.gobj", *mesh);

    int threw = 0;
    try { ModelSplatsHost bad(std::vector<float>{ 0, 0, 0 }, std::vector<float>{ 1, 2, 3 }, std::vector<float>{ 1, 1 }, { 1 }, { 1, 0, 0, 0 }); }
    catch (const std::runtime_error&) { threw++; }
    try {
        std::istringstream two("v 0 0 0\nsh 1 2 3\ns 1 1 1\na 1\nr 1 0 0 0\nv 0 0 0\nsh 1 2 3 4 5 6\ns 1 1 1\na 1\nr 1 0 0 0\n");
        readSplats(two);
    } catch (const std::runtime_error& e) { if (std::string(e.what()) == "Inconsistent SH degree!") threw++; }
    try {
        std::istringstream five("v 0 0 0\nf 1 1 1 1 1\n");
        parseObj(five);
    } catch (const std::runtime_error& e) { if (std::string(e.what()).rfind("Unexpected vertex count in face list!", 0) == 0) threw++; }
    try { loadSplats(dir + "/does_not_exist.gobj"); } catch (const std::runtime_error&) { threw++; }
    if (threw != 4) return 7;
    {   // a short / malformed fixed-width line still yields its 3 / 1 / 4 values, zero from the first failed extraction on
        std::istringstream shortLine("v 1 2\nsh 1 2 3\ns 1 nan 3\na\nr 1 0 0 0 9\n");
        auto m = readSplats(shortLine);
        if (m->count != 1 || m->locations[0] != 1.0f || m->locations[1] != 2.0f || m->locations[2] != 0.0f) return 8;
        if (m->scales[0] != 1.0f || m->scales[1] != 0.0f || m->scales[2] != 0.0f || m->opacities[0] != 0.0f || m->rotations[0] != 1.0f) return 8;
    }
    {   // Camera::getCameras / getPreviewCamera (src/Camera.cpp:33-74): written out for the Python mirror to compare with
        Project pr;
        pr.sphere1.count = 5; pr.sphere1.rotX = 40.0f; pr.sphere1.rotY = -15.0f;
        pr.sphere2.count = 3; pr.sphere2.distance = 6.0f; pr.sphere2.fovDeg = 30.0f; pr.sphere2.rotX = 200.0f;
        pr.previewTimer = 2.5f; pr.previewFreeRotY = 30.0f; pr.previewFreeDistance = 7.0f; pr.previewFreeFovDeg = 45.0f;
        FILE* f = fopen((dir + "/cpp_cameras.txt").c_str(), "w");
        if (!f) return 9;
        auto put = [f](const Camera& c) { fprintf(f, "%.9g %.9g %.9g %.9g\n", c.location[0], c.location[1], c.location[2], c.fovDegY); };
        if (Camera::getCamerasCount(pr) != 8) return 9;
        for (const Camera& c : Camera::getCameras(pr)) put(c);
        put(Camera::getPreviewCamera(pr));                          // free camera, orbiting
        pr.previewFreeOrbit = false; put(Camera::getPreviewCamera(pr));
        pr.previewTruth = true; pr.previewTruthIndex = 6; put(Camera::getPreviewCamera(pr));
        pr.previewTruthIndex = 8;
        bool past_end = false;
        try { Camera::getPreviewCamera(pr); } catch (const std::out_of_range&) { past_end = true; }
        fclose(f);
        if (!past_end) return 9;
    }
    {   // settings.json (UiFrame::saveSettings / loadSettings): <dir>/py_settings.json was written by io.py —
        //   <dir>/cpp_settings.json = load(py_settings.json) saved again   (must equal io.py's own load -> save, byte for byte)
        //   <dir>/cpp_made.json     = a project built here                   (io.py must read the same values back)
        Project loaded;
        loaded.iterations = 77; loaded.pathModel = "stale";       // WITH_DEFAULT: whatever the object held is replaced
        loadSettings(dir + "/py_settings.json", loaded);
        saveSettings(dir + "/cpp_settings.json", loaded);
        Project made;
        made.sphere2.count = 0; made.sphere2.fovDeg = 30.0f;
        made.lrSh = 0.1f; made.lrLocation = 1.0e-7f; made.paramScaleMax = 123456.0f; made.previewTimer = 1.0e22f; made.previewFreeRotY = -0.0f;
        made.iterations = 3491; made.previewTruth = true; made.previewFreeOrbit = false; made.renderResX = 4096;
        made.pathModel = "C:\\models\\a \"quoted\" \xc3\xa4.obj"; made.perspective = "layout2|name=a;caption=\tb\n|";
        saveSettings(dir + "/cpp_made.json", made);
        // a file that holds two keys: everything else takes a fresh Project's defaults; unknown keys are ignored; the last duplicate wins
        std::istringstream two("{ \"lrScale\": 0.5, \"sphere2\": {\"count\": 3}, \"noSuchKey\": [1, {\"a\": null}], \"lrScale\": 0.25 }");
        Project part;
        part.lrSh = 9.0f;
        readSettings(two, part);
        if (part.lrScale != 0.25f || part.sphere2.count != 3 || part.sphere2.distance != 10.0f || part.lrSh != 0.0001f || part.intervalDensify != 200) return 11;
        int refused = 0;
        for (const char* bad : { "{\"iterations\": \"many\"}", "{\"previewTruth\": 1}", "{\"sphere1\": 4}", "[1, 2]", "{\"lrSh\": 1.0", "{\"pathModel\": 3}" }) {
            std::istringstream is(bad);
            Project q;
            try { readSettings(is, q); } catch (const std::runtime_error&) { refused++; }
        }
        try { Project q; loadSettings(dir + "/does_not_exist.json", q); } catch (const std::runtime_error&) { refused++; }
        if (refused != 7) return 12;
        if (jsonNumber(0.1) != "0.1" || jsonNumber(2.0) != "2.0" || jsonNumber(1e22) != "1e+22" || jsonNumber(0.0001) != "0.0001" || jsonNumber(0.00001) != "1e-05" ||
            jsonNumber(1e16) != "1e+16" || jsonNumber(123456789012345.0) != "123456789012345.0" || jsonNumber((double)0.3f) != "0.30000001192092896") return 13;
    }
    {   // jsonNumber over <dir>/floats.bin (float32 values chosen by the Python side) -> <dir>/floats.txt, one number per line
        std::ifstream in(dir + "/floats.bin", std::ios::binary);
        if (in) {
            std::ofstream out(dir + "/floats.txt");
            float f;
            while (in.read(reinterpret_cast<char*>(&f), sizeof f)) out << jsonNumber((double)f) << '\n';
        }
    }
    {   // the auto-train loop (UiFrame::update, src/ui/UiFrame.cpp:266-298) against a trainer that only counts
        struct Counting {
            int trains = 0, densifies = 0, captures = 0, cameras = 0;
            void train(Project& p, bool densify) { p.iterations++; trains++; densifies += densify; }
            void captureTruths(const std::vector<Camera>& c, const std::vector<std::vector<uint32_t>>& w, const std::vector<std::vector<uint32_t>>& b) {
                captures++; cameras = (int)c.size();
                if (w.size() != c.size() || b.size() != c.size()) std::abort();
            }
        } counting;
        Project pr;
        pr.sphere1.count = 3; pr.sphere2.count = 0; pr.intervalCapture = 4; pr.intervalDensify = 6;
        unsigned lcg = 12345u;
        AutoTrainer<Counting, Project> loop(counting, pr,
            [](const std::vector<Camera>& cams, AutoTrainer<Counting, Project>::Frames& w, AutoTrainer<Counting, Project>::Frames& b) { w.assign(cams.size(), {}); b.assign(cams.size(), {}); },
            [&lcg] { lcg = lcg * 1664525u + 1013904223u; return (float)(lcg >> 8) / 16777216.0f; });
        int ran = 0;
        for (int tick = 0; tick < 400; tick++) ran += loop.update(0.0125f);      // idle events 12.5 ms apart: one iteration per event
        if (ran != 400 || pr.iterations != 400 || counting.trains != 400) return 10;
        if (counting.densifies != 67 || counting.captures != 100 || counting.cameras != 3) return 10;   // iterations 0, 6, ... 396 and 0, 4, ... 396
        if (!(pr.sphere1.rotX >= 0.0f && pr.sphere1.rotX < 360.0f) || pr.sphere1.rotX == 0.0f) return 10;
        ran = 0;
        for (int tick = 0; tick < 1000; tick++) ran += loop.update(0.001f);       // events 1 ms apart: the budget allows one iteration per ten
        if (ran < 95 || ran > 100) return 11;
        loop.autoTraining = false;
        if (loop.update(1.0f) || pr.previewTimer < 6.9f) return 11;                // the preview timer runs on (5 s + 1 s + 1 s)
    }
    printf("extras ok\n");
    return 0;
}
