// shim_step.cpp — drives include/gsplat_shim.hpp the way the reference's UI drives Trainer
// (src/ui/UiFrame.cpp:137-160, :266-298): build a host model, hand it to the trainer, capture truths,
// train, read the model back.  Inputs come from a binary file written by tests/test_gpu_shim.py; outputs
// go to another one, which the test compares bit-for-bit with the Python mirror's run.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "gsplat_shim.hpp"

using namespace gsplat_shim;

template <class T> static std::vector<T> rd(FILE* f, size_t n) {
    std::vector<T> v(n);
    if (n && fread(v.data(), sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(3); }
    return v;
}

int main(int argc, char** argv) {
    if (argc != 3) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    int hdr[6];  // P, M, W, H, cameras, steps
    if (fread(hdr, 4, 6, f) != 6) return 3;
    const int P = hdr[0], M = hdr[1], W = hdr[2], H = hdr[3], C = hdr[4], steps = hdr[5];
    auto loc = rd<float>(f, 3 * (size_t)P), sh = rd<float>(f, 3 * (size_t)M * P), scale = rd<float>(f, 3 * (size_t)P),
         opac = rd<float>(f, P), rot = rd<float>(f, 4 * (size_t)P);
    auto camf = rd<float>(f, 7 * (size_t)C);  // per camera: location, target, fovDegY
    std::vector<std::vector<uint32_t>> fw, fb;
    for (int c = 0; c < C; c++) fw.push_back(rd<uint32_t>(f, (size_t)W * H));
    for (int c = 0; c < C; c++) fb.push_back(rd<uint32_t>(f, (size_t)W * H));
    fclose(f);
    try {
        Trainer trainer(W, H);
        Trainer* tp = &trainer;
        Project project;
        bool threw = false;
        try { tp->train(project, false); } catch (const std::runtime_error&) { threw = true; }  // src/Trainer.cu:253
        if (!threw || project.iterations != 0) return 4;  // the reference throws before it counts the iteration
        ModelSplatsHost host(loc, sh, scale, opac, rot);
        delete tp->model;                              // the reference idiom verbatim, src/ui/UiFrame.cpp:157-158
        tp->model = new ModelSplatsDevice(host);
        std::vector<Camera> cameras;
        for (int c = 0; c < C; c++) cameras.emplace_back(&camf[7 * c], &camf[7 * c + 3], camf[7 * c + 6]);
        tp->captureTruths(cameras, fw, fb);
        if ((int)tp->truthCameras.size() != C) return 5;
        for (int s = 0; s < steps; s++) tp->train(project, false);          // src/ui/UiFrame.cpp:288
        if (project.iterations != steps) return 6;
        ModelSplatsHost back(*tp->model);
        std::vector<uint32_t> frame((size_t)W * H);
        tp->render(frame.data(), W, H, project.previewSplatScale, cameras[0]);  // src/ui/UiPanelViewOutput.cpp:52-60
        FILE* o = fopen(argv[2], "wb");
        fwrite(&back.count, 4, 1, o);
        fwrite(tp->truthViewsW.data(), sizeof(gs_view), tp->truthViewsW.size(), o);  // the pass parameters Camera::pass derived
        fwrite(tp->truthViewsB.data(), sizeof(gs_view), tp->truthViewsB.size(), o);
        fwrite(back.locations, 4, 3 * (size_t)back.count, o);
        fwrite(back.opacities, 4, (size_t)back.count, o);
        fwrite(frame.data(), 4, frame.size(), o);
        fclose(o);
        // Checkpoint / resume of an Adam run through the shim (gs_trainer_adam_state / gs_trainer_set_adam_state), and the camera
        // every render caller of the reference builds (Camera::getPreviewCamera(*project), src/ui/UiPanelViewOutput.cpp:52-60)
        {
            Trainer a(W, H);
            a.updateRule = GS_UPDATE_ADAM;
            delete a.model;
            a.model = new ModelSplatsDevice(host);
            a.captureTruths(cameras, fw, fb);
            Project pa;
            pa.lrLocation = 1e-3f; pa.lrSh = 2e-3f;
            std::vector<float> m1, m2;
            if (a.adamState(m1, m2) != 0 || !m1.empty()) return 7;      // no Adam step yet
            for (int s = 0; s < 2; s++) a.train(pa, false);
            ModelSplatsHost saved(*a.model);
            const int taken = a.adamState(m1, m2);
            if (taken != 2 || m1.size() != m2.size() || m1.empty()) return 7;
            for (int s = 0; s < 2; s++) a.train(pa, false);
            ModelSplatsHost straight(*a.model);
            Trainer b(W, H);
            b.updateRule = GS_UPDATE_ADAM;
            delete b.model;
            b.model = new ModelSplatsDevice(saved);
            b.captureTruths(cameras, fw, fb);
            b.setAdamState(m1, m2, taken);
            Project pb = pa;
            for (int s = 0; s < 2; s++) b.train(pb, false);
            ModelSplatsHost resumed(*b.model);
            if (resumed.count != straight.count || memcmp(resumed.locations, straight.locations, 12 * (size_t)straight.count) != 0 ||
                memcmp(resumed.shs, straight.shs, 12 * (size_t)M * straight.count) != 0) return 8;    // bit for bit
            Project pv;
            pv.previewTimer = 1.0f;
            std::vector<uint32_t> pf((size_t)W * H);
            b.render(pf.data(), W, H, pv.previewSplatScale, Camera::getPreviewCamera(pv));
            size_t lit = 0;
            for (uint32_t px : pf) lit += (px & 0xFFFFFFu) != 0;
            if (lit == 0) return 9;                                      // the free camera at distance 10 sees the scene
        }
    } catch (const std::exception& e) {
        fprintf(stderr, "shim_step: %s\n", e.what());
        return 1;
    }
    return 0;
}
