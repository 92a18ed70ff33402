// shim_step.cpp — drives include/gsplat_shim.hpp the way the reference's UI drives Trainer
// (src/ui/UiFrame.cpp:137-160, :266-298): build a host model, hand it to the trainer, capture truths,
// train, read the model back.  Inputs come from a binary file written by tests/test_gpu_shim.py; outputs
// go to another one, which the test compares bit-for-bit with the Python mirror's run.
#include <cstdio>
#include <cstdlib>
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
    auto views = rd<gs_view>(f, 2 * (size_t)C);
    std::vector<std::vector<uint32_t>> fw, fb;
    for (int c = 0; c < C; c++) fw.push_back(rd<uint32_t>(f, (size_t)W * H));
    for (int c = 0; c < C; c++) fb.push_back(rd<uint32_t>(f, (size_t)W * H));
    fclose(f);
    try {
        Trainer trainer(W, H);
        bool threw = false;
        gs_hyper hyper;
        check(gs_hyper_defaults(&hyper));
        try { trainer.train(hyper, false); } catch (const std::runtime_error&) { threw = true; }  // src/Trainer.cu:253
        if (!threw) return 4;
        ModelSplatsHost host(loc, sh, scale, opac, rot);
        delete trainer.model;                          // the reference idiom, src/ui/UiFrame.cpp:157-158
        trainer.model = new ModelSplatsDevice(host);
        trainer.adoptModel();
        trainer.captureTruths(std::vector<gs_view>(views.begin(), views.begin() + C), std::vector<gs_view>(views.begin() + C, views.end()), fw, fb);
        std::vector<float> losses;
        for (int s = 0; s < steps; s++) losses.push_back(trainer.train(hyper, false).loss);
        ModelSplatsHost back(*trainer.model);
        std::vector<uint32_t> frame((size_t)W * H);
        trainer.render(frame.data(), W, H, 1.0f, views[C]);  // black-background pass of camera 0
        FILE* o = fopen(argv[2], "wb");
        fwrite(&back.count, 4, 1, o);
        fwrite(losses.data(), 4, losses.size(), o);
        fwrite(back.locations, 4, 3 * (size_t)back.count, o);
        fwrite(back.opacities, 4, (size_t)back.count, o);
        fwrite(frame.data(), 4, frame.size(), o);
        fclose(o);
    } catch (const std::exception& e) {
        fprintf(stderr, "shim_step: %s\n", e.what());
        return 1;
    }
    return 0;
}
