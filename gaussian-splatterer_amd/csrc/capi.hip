// capi.hip — the C-ABI of include/gsplat.h: model, trainer, rasterizer seam, RCCL communicator.
// Host orchestration only; every arithmetic step is a HIP kernel in k_*.hip (densify / prune
// included: k_densify.hip — the reference does that one on the CPU).  There is no CPU fallback:
// without a HIP device the entry points fail with GS_ERR_NO_DEVICE / GS_ERR_HIP.
#include <dlfcn.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "gs_internal.h"

namespace gs {

static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// Switches of a trainer (gs_trainer_set_option).  gs_set_option edits the process-wide DEFAULTS: a trainer copies them when
// it is created and is not affected by later changes; the rasterizer seam (which has no trainer) reads `cull` from them.
struct Options {
    int cull = 1;
    int share = 1;       // passes with bit-identical cameras share projection, lists and the forward blend
    int arena = 0;       // initial binning-arena entries per camera (0 = default max(2^20, 16 P))
    int fuse = 1;        // gs_trainer_step without densify: one backward per camera pair (the step's `var` has no reader)
    int debug_sync = 0;  // wait and check for errors after every stage, like the reference's debug=true rasterizer calls
    int sh_fp16 = 0;     // the projection reads SH coefficients from a half-precision copy (BASELINE cfg5); fp32 master and gradients
    int long_sort = -1;  // k_tile_sort_long launch: -1 by the longest-list hint (default), 0 never (lists take the global-scratch path), 1 always
    int sort_grids = -1; // test hook: >= 0 replaces the device's hint for the sort grids: small_first | mid_grid << 16 (in tiles)
    int fuse_update = 1;  // a step with no collective between gradients and update applies the update inside the per-splat reduction (no update launch)
    int xchg_overlap = 1; // compact exchange: the all-reduce of the geometry planes runs on a second stream beside the all-gather (0: one after the other)
    int list_cut = 1;     // tile lists cut at the depth the previous step's forward stopped looking (Dims::cut): 0 never, 1 where it pays (dense scenes)
    int list_cut_min_avg = 384;  // ... i.e. from this many entries per tile on average (the previous step's count); test hook
    int list_cut_margin = 64;    // entries kept behind the last one the forward looked at; test hook (negative: cuts that must be found wrong)
    int row_marks = -1;   // gradient rows only for evaluated entries (row_epoch marks): -1 per camera by its longest tile list (from 1024 entries), 0 never, 1 always
    int reuse_masks = 1;  // the backward reuses the forward's per-(tile sub-block, wave) block ballots; 0: it runs the block test itself (same bits)
    int roctx = 0;        // roctx range around every stage of a step (rocprofv3 --marker-trace names them); default from the environment: GS_ROCTX=1
};
static Options g_defaults;
// returns false for an unknown name
static bool set_option(Options& o, const char* name, int value) {
    if (strcmp(name, "cull") == 0) { o.cull = value != 0; return true; }
    if (strcmp(name, "share_camera_passes") == 0) { o.share = value != 0; return true; }
    if (strcmp(name, "fuse_camera_passes") == 0) { o.fuse = value != 0; return true; }
    if (strcmp(name, "arena_entries") == 0) { o.arena = value > 0 ? value : 0; return true; }
    if (strcmp(name, "debug_sync") == 0) { o.debug_sync = value != 0; return true; }
    if (strcmp(name, "sh_fp16") == 0) { o.sh_fp16 = value != 0; return true; }
    if (strcmp(name, "long_list_sort_launch") == 0) { o.long_sort = value < 0 ? -1 : (value != 0); return true; }
    if (strcmp(name, "debug_sort_grids") == 0) { o.sort_grids = value < 0 ? -1 : value; return true; }
    if (strcmp(name, "exchange_overlap") == 0) { o.xchg_overlap = value != 0; return true; }
    if (strcmp(name, "fuse_update") == 0) { o.fuse_update = value != 0; return true; }
    if (strcmp(name, "roctx") == 0) { o.roctx = value != 0; return true; }
    if (strcmp(name, "reuse_hit_masks") == 0) { o.reuse_masks = value != 0; return true; }
    if (strcmp(name, "list_cut") == 0) { o.list_cut = value != 0; return true; }
    if (strcmp(name, "list_cut_min_avg") == 0) { o.list_cut_min_avg = value < 0 ? 0 : value; return true; }
    if (strcmp(name, "list_cut_margin") == 0) { o.list_cut_margin = value; return true; }
    if (strcmp(name, "row_marks") == 0) { o.row_marks = value < 0 ? -1 : (value != 0); return true; }
    return false;
}

// grow-only device buffer
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    uint64_t generation = 0;   // advanced by every (re)allocation and release: what "is this still the buffer I initialised?" must compare —
                               // the allocator may hand the SAME address back for a larger block after the free
    int ensure(size_t bytes) {
        if (bytes <= cap) return GS_OK;
        if (p) { GS_HIP(hipFree(p)); p = nullptr; cap = 0; }
        generation++;
        bytes = round_up_sz(bytes, 256);
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) { p = nullptr; set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); return GS_ERR_OUT_OF_MEMORY; }
        cap = bytes;
        return GS_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; generation++; }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

static int effective_degree(int sh_degree, int M, int* D_out) {
    const int D = sh_degree > 3 ? 3 : (sh_degree < 0 ? 0 : sh_degree);  // the rasterizer treats every degree > 2 as 3
    if (M < 1 || (D + 1) * (D + 1) > M) {
        set_error("SH degree %d needs %d coefficients but the model carries %d", sh_degree, (D + 1) * (D + 1), M);
        return GS_ERR_DIMENSIONS;
    }
    *D_out = D;
    return GS_OK;
}

static Dims make_dims(int P, int Pa, int D, int M, int W, int H, int V, uint32_t Rcap, float mod, int VG = -1, int cull = -1) {
    Dims d;
    d.P = P; d.Pa = Pa; d.D = D; d.M = M; d.W = W; d.H = H; d.N = W * H;
    d.gx = (W + TILE - 1) / TILE; d.gy = (H + TILE - 1) / TILE; d.T = d.gx * d.gy;
    d.sgx = (d.gx + STILE - 1) / STILE; d.sgy = (d.gy + STILE - 1) / STILE; d.NST = d.sgx * d.sgy;
    d.V = V; d.VG = VG < 0 ? V : VG; d.Rcap = Rcap; d.mod = mod; d.cull = cull < 0 ? g_defaults.cull : cull; d.long_sort = 1; d.mid_sort = 1; d.small_first = 0; d.mid_grid = 0;
    d.epoch = 1; d.marks_min_list = 0; d.cut = 0; d.cut_track = 0; d.cut_margin = 0;
    return d;
}

// Per-view scratch owned by the library (trainer / preview).  One allocation per array family,
// grow-only, sized for V views at once: with 288 GB of HBM every view of a step keeps its own
// state and each stage is ONE launch over all views.
// One device block carries everything that describes the passes of a step: the per-pass view structs, the camera
// of every geometry group, and the pass <-> group maps.  One host-to-device copy per step.
static size_t view_block_bytes(int V) { return (size_t)V * 2 * sizeof(gs_view) + ((size_t)6 * V + 1) * sizeof(int); }
// int offset of the backward work items {group, pass a, pass b | -1} inside the map area
static size_t view_block_items_offset(int V) { return (size_t)V * 2 * sizeof(gs_view) + ((size_t)3 * V + 1) * sizeof(int); }
static void set_view_block_pointers(Scratch& s, char* base, int V) {
    s.views = reinterpret_cast<const gs_view*>(base);
    s.gviews = s.views + V;
    const int* m = reinterpret_cast<const int*>(s.gviews + V);
    s.view_group = m; s.group_first = m + V; s.group_views = m + 2 * V + 1;
}
// Groups the passes by bit-identical camera (view, projview, campos, tan_fov*): the reference runs a white- and a
// black-background pass per camera (src/Trainer.cu:311-318) whose projection, tile lists and blend are identical.
static int build_view_block(const gs_view* views, int V, bool share, std::vector<char>& out, int* n_pairs = nullptr, int* n_singles = nullptr) {
    out.assign(view_block_bytes(V), 0);
    gs_view* pv = reinterpret_cast<gs_view*>(out.data());
    gs_view* gv = pv + V;
    int* vg = reinterpret_cast<int*>(gv + V);
    int* gfirst = vg + V;
    int* glist = gfirst + V + 1;
    int VG = 0;
    for (int v = 0; v < V; v++) {
        pv[v] = views[v];
        int g = -1;
        if (share)
            for (int k = 0; k < VG && g < 0; k++)
                if (memcmp(&gv[k], &views[v], offsetof(gs_view, bg)) == 0) g = k;
        if (g < 0) { g = VG++; gv[g] = views[v]; }
        vg[v] = g;
    }
    int pos = 0;
    for (int g = 0; g < VG; g++) {
        gfirst[g] = pos;
        for (int v = 0; v < V; v++) if (vg[v] == g) glist[pos++] = v;
    }
    gfirst[VG] = pos;
    // backward work items: the passes of a camera two at a time (pairs first, then leftovers)
    int* items = reinterpret_cast<int*>(out.data() + view_block_items_offset(V));
    int n2 = 0, n1 = 0;
    for (int g = 0; g < VG; g++)
        for (int k = gfirst[g]; k + 1 < gfirst[g + 1]; k += 2) { int* it = items + 3 * n2++; it[0] = g; it[1] = glist[k]; it[2] = glist[k + 1]; }
    for (int g = 0; g < VG; g++)
        if ((gfirst[g + 1] - gfirst[g]) & 1) { int* it = items + 3 * (n2 + n1++); it[0] = g; it[1] = glist[gfirst[g + 1] - 1]; it[2] = -1; }
    if (n_pairs) *n_pairs = n2;
    if (n_singles) *n_singles = n1;
    return VG;
}

struct ScratchSet {
    DevBuf views, geom, tiles, offsets, tloss, torder, zero_block, wghist, colscan, coarse_count, coarse_end, tile_count, tile_end, clist, cdepth, ids, plist, slist, G, rowmark, hmask, color, finalT, ncontrib, scan_tmp, sgrads, shjac, meancopy, zcut, szcut;
    uint64_t rowmark_cleared = ~0ull;  // the generation of `rowmark` that has been zeroed (a new allocation holds stale bytes: the caller clears it and restarts the epochs)
    int V = 0, Pa = 0, T = 0, N = 0, NST = 0;
    uint32_t Rcap = 0;
    Scratch s{};
    size_t zero_bytes = 0;

    int ensure(int P, int V_, int W, int H, uint32_t Rcap_, bool want_splat_grads = false) {
        Pa = std::max(64, round_up(P, 64));
        V = V_; N = W * H;
        T = ((W + TILE - 1) / TILE) * ((H + TILE - 1) / TILE);
        if (Rcap_ >= (1u << 31) - 64u) { set_error("binning arena: %u entries per view exceed the 2^31 the index arithmetic holds", Rcap_); return GS_ERR_OUT_OF_MEMORY; }
        Rcap = (std::max<uint32_t>(Rcap_, 1024) + 63u) & ~63u;  // multiple of 64: every view's slice of G (36-byte rows) stays 8-byte aligned for the sort keys
        const size_t v = (size_t)std::max(V, 1);
        GS_TRY(views.ensure(view_block_bytes((int)v)));
        GS_TRY(geom.ensure(v * Pa * sizeof(GeomRec)));
        GS_TRY(tiles.ensure(v * Pa * 4));
        GS_TRY(offsets.ensure(v * Pa * 4));
        {
            const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
            NST = ((gx + STILE - 1) / STILE) * ((gy + STILE - 1) / STILE);
        }
        // flags | loss totals (every word is rewritten by the kernels of a step: no memset)
        zero_bytes = v * 16 + v * 4 + v * 8;  // flags[4] per group | loss totals | sort marks[2] per group
        GS_TRY(zero_block.ensure(zero_bytes));
        GS_TRY(wghist.ensure(v * splat_blocks(Pa) * NST * 4));
        GS_TRY(coarse_count.ensure(v * NST * 4));
        GS_TRY(colscan.ensure(std::max<size_t>(4, colscan_partial_words(Pa, NST, (int)v) * 4)));
        GS_TRY(coarse_end.ensure(v * NST * 4));
        GS_TRY(tile_count.ensure(v * T * 4));
        GS_TRY(tile_end.ensure(v * T * 4));
        GS_TRY(tloss.ensure(v * T * 4));
        GS_TRY(torder.ensure(v * T * 4));
        GS_TRY(clist.ensure(v * Rcap * 16));
        GS_TRY(cdepth.ensure(v * Rcap * 4));
        GS_TRY(ids.ensure(v * Rcap * 4));
        GS_TRY(plist.ensure(v * Rcap * 4));
        GS_TRY(slist.ensure(v * Rcap * 4));
        GS_TRY(G.ensure(v * Rcap * G_STRIDE * 4));
        GS_TRY(rowmark.ensure(want_splat_grads ? 64 + v * Rcap + 64 : 128));  // [64 zero bytes: Scratch::zero_row][marks][slack for the marks read one trip ahead]
        GS_TRY(hmask.ensure(want_splat_grads ? v * hit_mask_words(Rcap, T) * 4 * sizeof(unsigned long long) : 8));
        GS_TRY(color.ensure(v * 3 * N * 4));
        GS_TRY(finalT.ensure(v * N * 4));
        GS_TRY(ncontrib.ensure(v * N * 4));
        // [block_sums: v x splat_blocks(Pa)] [partials of the three-phase scan, used only past g_scan_single_max items]
        GS_TRY(scan_tmp.ensure((v * splat_blocks(Pa) + scan_partials_count(std::max(T, NST), (int)v) + 64) * 4));
        s.block_sums = scan_tmp.as<uint32_t>();
        if (want_splat_grads) { GS_TRY(sgrads.ensure(v * Pa * 64)); GS_TRY(shjac.ensure(v * Pa * 48)); GS_TRY(meancopy.ensure((size_t)3 * Pa * 4)); GS_TRY(zcut.ensure(v * T * 4)); GS_TRY(szcut.ensure(v * NST * 4)); }
        s.splat_grads = sgrads.as<float>();
        s.sh_jac = want_splat_grads ? shjac.as<float>() : nullptr;
        s.mean_copy = want_splat_grads ? meancopy.as<float>() : nullptr;
        s.tile_zcut = want_splat_grads ? zcut.as<uint32_t>() : nullptr;
        s.stile_zcut = want_splat_grads ? szcut.as<uint32_t>() : nullptr;
        set_view_block_pointers(s, views.as<char>(), (int)v);
        s.geom = geom.as<GeomRec>();
        s.tiles_touched = tiles.as<uint32_t>();
        s.point_offsets = offsets.as<uint32_t>();
        s.wg_hist = wghist.as<uint32_t>();
        s.coarse_count = coarse_count.as<uint32_t>();
        s.colscan_partial = colscan_partial_words(Pa, NST, (int)v) ? colscan.as<uint32_t>() : nullptr;
        s.flags = zero_block.as<uint32_t>();
        s.loss_total = reinterpret_cast<float*>(s.flags + v * 4);
        s.sort_marks = s.flags + v * 5;
        s.loss = tloss.as<float>();
        s.tile_order = torder.as<uint32_t>();
        s.coarse_end = coarse_end.as<uint32_t>();
        s.tile_count = tile_count.as<uint32_t>();
        s.tile_end = tile_end.as<uint32_t>();
        s.coarse_list = clist.as<uint4>();
        s.coarse_depth = cdepth.as<uint32_t>();
        s.id_of_slot = ids.as<uint32_t>();
        s.point_list = plist.as<uint32_t>();
        s.slot_list = slist.as<uint32_t>();
        s.G = G.as<float>();
        s.zero_row = rowmark.as<float>();
        s.row_epoch = rowmark.as<uint8_t>() + 64;
        s.hit_masks = want_splat_grads ? hmask.as<unsigned long long>() : nullptr;  // only a trainer runs a backward behind the forward
        s.out_color = color.as<float>();
        s.final_T = finalT.as<float>();
        s.n_contrib = ncontrib.as<uint32_t>();
        s.truth = nullptr;
        s.dL_dpix = nullptr;
        return GS_OK;
    }
    void release() {
        for (DevBuf* b : { &views, &geom, &tiles, &offsets, &tloss, &torder, &zero_block, &wghist, &colscan, &coarse_count, &coarse_end, &tile_count, &tile_end, &clist, &cdepth, &ids, &plist, &slist, &G, &rowmark, &hmask, &color,
                           &finalT, &ncontrib, &scan_tmp, &sgrads, &shjac, &meancopy, &zcut, &szcut })
            b->release();
    }
};

// projection + everything that does not need the binning arena: the column scan of the count matrix and, in the same
// launch, the block prefixes of the offsets scan (entry count and overflow bit land in flags).  Runs for P == 0 too
// (the scans then define empty lists).
static int stage_project(const Dims& d, const float* params, const Scratch& s, hipStream_t st) {
    if (d.NST > MAX_SUPER_TILES) {
        set_error("image of %dx%d has %d super-tiles; this build bins up to %d (8192x8192, 16384x4096, ...)", d.W, d.H, d.NST, MAX_SUPER_TILES);
        return GS_ERR_INVALID_ARGUMENT;
    }
    GS_TRY(launch_preprocess(d, params, s, st));
    GS_TRY(launch_coarse_colscan(d, s, st));
    return GS_OK;
}
// coarse scatter (finishes the offsets scan, scans the super-tile totals), per-tile counts, then ONE launch for the
// per-tile segments, the tile scan and the longest-first tile order
static int stage_bin(const Dims& d, const Scratch& s, hipStream_t st) {
    GS_TRY(launch_coarse_scatter(d, s, st));
    GS_TRY(launch_tile_count(d, s, st));
    GS_TRY(launch_tile_scatter(d, s, s.block_sums + (size_t)std::max(d.VG, 1) * splat_blocks(d.Pa), st));
    return GS_OK;
}
static int stage_bin_render(const Dims& d, const Scratch& s, hipStream_t st) {
    GS_TRY(stage_bin(d, s, st));
    GS_TRY(launch_tile_build_sort(d, s, st));
    GS_TRY(launch_render_forward(d, s, st));
    return GS_OK;
}

}  // namespace gs

using namespace gs;

// =============================================================================================
// misc
// =============================================================================================
extern "C" const char* gs_last_error(void) { return g_err; }
extern "C" const char* gs_status_string(int s) {
    switch (s) {
        case GS_OK: return "ok";
        case GS_ERR_INVALID_ARGUMENT: return "invalid argument";
        case GS_ERR_HIP: return "HIP runtime error";
        case GS_ERR_NO_TRUTH: return "Can't run training iteration, no truth data available!";
        case GS_ERR_CAPACITY: return "Model ran out of capacity!";
        case GS_ERR_BOUNDS: return "Can't copy splat in model, incorrect bounds and/or no capacity!";
        case GS_ERR_DIMENSIONS: return "Inconsistent feature dimensions supplied when creating a host model!";
        case GS_ERR_OUT_OF_MEMORY: return "out of device memory";
        case GS_ERR_NO_MODEL: return "trainer has no model";
        case GS_ERR_NO_DEVICE: return "no HIP device (this library has no CPU fallback)";
        case GS_ERR_COLLECTIVE: return "a data-parallel collective failed on this rank: end the process so that the launcher ends the peers";
        default: return "internal error";
    }
}
extern "C" int gs_version(int* a, int* b, int* c) {
    if (a) *a = GS_VERSION_MAJOR;
    if (b) *b = GS_VERSION_MINOR;
    if (c) *c = GS_VERSION_PATCH;
    return GS_OK;
}
extern "C" int gs_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
static int require_device() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
        set_error("no HIP device visible; libgsplat_mi355 has no CPU path");
        return GS_ERR_NO_DEVICE;
    }
    return GS_OK;
}
extern "C" int gs_device_malloc(void** p, size_t bytes) {
    if (!p) return GS_ERR_INVALID_ARGUMENT;
    GS_TRY(require_device());
    GS_HIP(hipMalloc(p, bytes ? bytes : 1));
    return GS_OK;
}
extern "C" int gs_device_free(void* p) { GS_HIP(hipFree(p)); return GS_OK; }
extern "C" int gs_memcpy_h2d(void* d, const void* h, size_t n) { if (n) GS_HIP(hipMemcpy(d, h, n, hipMemcpyHostToDevice)); return GS_OK; }
extern "C" int gs_memcpy_d2h(void* h, const void* d, size_t n) { if (n) GS_HIP(hipMemcpy(h, d, n, hipMemcpyDeviceToHost)); return GS_OK; }
extern "C" int gs_memset_d(void* d, int v, size_t n) { if (n) GS_HIP(hipMemset(d, v, n)); return GS_OK; }
extern "C" int gs_device_synchronize(void) { GS_HIP(hipDeviceSynchronize()); return GS_OK; }

extern "C" int gs_set_option(const char* name, int value) {
    if (!name) return GS_ERR_INVALID_ARGUMENT;
    if (set_option(g_defaults, name, value)) return GS_OK;
    if (strcmp(name, "scan_single_max") == 0) { gs::g_scan_single_max = value > 0 ? value : (1 << 16); return GS_OK; }
    set_error("gs_set_option: unknown option '%s'", name);
    return GS_ERR_INVALID_ARGUMENT;
}

// Diagnostic: the render-backward kernel's 9-value reduce-scatter over the 16-lane rows of one wave.
// in_host[q*64 + lane], out_host[lane]: lane 2q of a row = that ROW's total of q (q < 8), lane 1 = its total of q = 8.
extern "C" int gs_debug_wave_reduce9(const float* in_host, float* out_host) {
    if (!in_host || !out_host) return GS_ERR_INVALID_ARGUMENT;
    GS_TRY(require_device());
    float *din = nullptr, *dout = nullptr;
    GS_HIP(hipMalloc((void**)&din, 9 * 64 * 4));
    GS_HIP(hipMalloc((void**)&dout, 64 * 4));
    GS_HIP(hipMemcpy(din, in_host, 9 * 64 * 4, hipMemcpyHostToDevice));
    int rc = launch_debug_reduce9(din, dout, 0);
    if (rc == GS_OK && hipMemcpy(out_host, dout, 64 * 4, hipMemcpyDeviceToHost) != hipSuccess) rc = GS_ERR_HIP;
    (void)hipFree(din); (void)hipFree(dout);
    return rc;
}

extern "C" int gs_debug_counters(unsigned long long out[8], int reset) {
    if (!out) return GS_ERR_INVALID_ARGUMENT;
    GS_TRY(require_device());
    return debug_counters(out, reset != 0);
}

static int g_copy_probe_best_form = -1;   // which form won the last gs_debug_hbm_copy_rate (diagnostic; gs_last_error text of a successful call is unchanged)
extern "C" int gs_debug_hbm_copy_rate(size_t bytes, int repeats, double* gbytes_per_s) {
    if (!gbytes_per_s || bytes < 16 || (bytes & 15) || repeats < 1 || repeats > 1000) { set_error("gs_debug_hbm_copy_rate: bad arguments"); return GS_ERR_INVALID_ARGUMENT; }
    GS_TRY(require_device());
    void *a = nullptr, *b = nullptr;
    hipStream_t st = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = GS_OK;
    double best_ms = 0.0;
    auto fail = [&](const char* what, hipError_t e) { set_error("gs_debug_hbm_copy_rate: %s: %s", what, hipGetErrorString(e)); rc = e == hipErrorOutOfMemory ? GS_ERR_OUT_OF_MEMORY : GS_ERR_HIP; };
    hipError_t e = hipMalloc(&a, bytes);
    if (e == hipSuccess) e = hipMalloc(&b, bytes);
    if (e != hipSuccess) fail("hipMalloc", e);
    if (rc == GS_OK && (e = hipMemset(a, 0x3c, bytes)) != hipSuccess) fail("hipMemset", e);
    if (rc == GS_OK && (e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking)) != hipSuccess) fail("hipStreamCreate", e);
    if (rc == GS_OK && ((e = hipEventCreate(&e0)) != hipSuccess || (e = hipEventCreate(&e1)) != hipSuccess)) fail("hipEventCreate", e);
    // sixteen forms of the copy (grid of 256 x 8 ... 64 workgroups, one or four float4 in flight per lane, plain or non-temporal accesses),
    // each launched `repeats` times behind one warm-up: the fastest single launch of all is the figure
    for (int form = 0; rc == GS_OK && form < 16; form++)
        for (int k = 0; rc == GS_OK && k <= repeats; k++) {   // k = 0: warm-up (first touch of the destination)
            (void)hipEventRecord(e0, st);
            rc = launch_copy_probe(a, b, bytes, form, st);
            (void)hipEventRecord(e1, st);
            if (rc == GS_OK && (e = hipEventSynchronize(e1)) != hipSuccess) fail("hipEventSynchronize", e);
            float ms = 0.0f;
            if (rc == GS_OK && (e = hipEventElapsedTime(&ms, e0, e1)) != hipSuccess) fail("hipEventElapsedTime", e);
            if (rc == GS_OK && k > 0 && ms > 0.0f && (best_ms == 0.0 || ms < best_ms)) { best_ms = ms; g_copy_probe_best_form = form; }
        }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (st) (void)hipStreamDestroy(st);
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    if (rc != GS_OK) return rc;
    *gbytes_per_s = best_ms > 0.0 ? 2.0 * (double)bytes / (best_ms * 1e-3) / 1e9 : 0.0;
    return GS_OK;
}

extern "C" int gs_debug_hbm_copy_form(void) { return g_copy_probe_best_form; }

extern "C" int gs_hyper_defaults(gs_hyper* h) {
    if (!h) return GS_ERR_INVALID_ARGUMENT;
    h->lr_location = 0.00005f; h->lr_sh = 0.0001f; h->lr_scale = 0.00002f; h->lr_opacity = 0.0001f; h->lr_rotation = 0.000025f;
    h->scale_max = 0.3f;
    h->cull_opacity = 0.005f; h->cull_size = 0.004f; h->densify_variance = 2.0f;
    h->split_size = 0.04f; h->split_distance = 1.5f; h->split_scale = 0.8f; h->clone_distance = 1.6f;
    h->update_rule = GS_UPDATE_SGD_CLAMP;
    h->adam_beta1 = 0.9f; h->adam_beta2 = 0.999f; h->adam_eps = 1e-15f;
    h->quat_layout = GS_QUAT_XYZW;
    return GS_OK;
}

// =============================================================================================
// model
// =============================================================================================
static int model_alloc(int capacity, int sh_degree, int sh_coeffs, int count, gs_model** out) {
    gs_model* m = new gs_model();
    m->capacity = capacity; m->sh_degree = sh_degree; m->sh_coeffs = sh_coeffs; m->count = count;
    if (hipGetDevice(&m->device) != hipSuccess) { delete m; set_error("hipGetDevice failed"); return GS_ERR_HIP; }
    m->Pa = std::max(64, round_up(count, 64));
    const size_t bytes = plane_buffer_floats(sh_coeffs, m->Pa) * sizeof(float);  // parameter planes + spare plane + chunk padding
    hipError_t e = hipMalloc((void**)&m->planes, bytes);
    if (e != hipSuccess) { delete m; set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); return GS_ERR_OUT_OF_MEMORY; }
    m->planes_bytes = bytes;
    e = hipMemset(m->planes, 0, bytes);
    if (e != hipSuccess) { (void)hipFree(m->planes); delete m; set_error("hipMemset failed: %s", hipGetErrorString(e)); return GS_ERR_HIP; }
    *out = m;
    return GS_OK;
}

extern "C" int gs_model_create(int capacity, int sh_degree, int sh_coeffs, int count, const float* loc, const float* sh,
                               const float* scale, const float* opac, const float* rot, gs_model** out) {
    if (!out || capacity < 0 || count < 0 || sh_coeffs < 1) { set_error("gs_model_create: bad arguments"); return GS_ERR_INVALID_ARGUMENT; }
    if (count > capacity) { set_error("Model ran out of capacity! (count %d > capacity %d)", count, capacity); return GS_ERR_CAPACITY; }
    if (count > 0 && (!loc || !sh || !scale || !opac || !rot)) {
        set_error("Inconsistent feature dimensions supplied when creating a host model! (null array)");
        return GS_ERR_DIMENSIONS;
    }
    GS_TRY(require_device());
    gs_model* m = nullptr;
    GS_TRY(model_alloc(capacity, sh_degree, sh_coeffs, count, &m));
    if (count > 0) {
        // stage the reference-layout arrays and transpose to planes on the device
        const size_t nf = (size_t)count * (11 + 3 * sh_coeffs);
        float* stage = nullptr;
        hipError_t e = hipMalloc((void**)&stage, nf * sizeof(float));
        if (e != hipSuccess) { gs_model_destroy(m); set_error("hipMalloc staging failed: %s", hipGetErrorString(e)); return GS_ERR_OUT_OF_MEMORY; }
        float* dl = stage; float* dsh = dl + 3 * (size_t)count; float* dsc = dsh + 3 * (size_t)sh_coeffs * count;
        float* dop = dsc + 3 * (size_t)count; float* dr = dop + count;
        int rc = GS_OK;
        auto cp = [&](float* d, const float* h, size_t n) { if (rc == GS_OK && hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice) != hipSuccess) rc = GS_ERR_HIP; };
        cp(dl, loc, 3 * (size_t)count); cp(dsh, sh, 3 * (size_t)sh_coeffs * count); cp(dsc, scale, 3 * (size_t)count);
        cp(dop, opac, count); cp(dr, rot, 4 * (size_t)count);
        if (rc == GS_OK) rc = launch_aos_to_soa(count, m->Pa, sh_coeffs, dl, dsh, dsc, dop, dr, m->planes, 0);
        if (rc == GS_OK && hipDeviceSynchronize() != hipSuccess) rc = GS_ERR_HIP;
        (void)hipFree(stage);
        if (rc != GS_OK) { gs_model_destroy(m); if (rc == GS_ERR_HIP) set_error("model upload failed: %s", hipGetErrorString(hipGetLastError())); return rc; }
    }
    *out = m;
    return GS_OK;
}

extern "C" int gs_model_clone(const gs_model* src, gs_model** out) {
    if (!src || !out) return GS_ERR_INVALID_ARGUMENT;
    gs_model* m = nullptr;
    GS_TRY(model_alloc(src->capacity, src->sh_degree, src->sh_coeffs, src->count, &m));
    const size_t bytes = (size_t)(11 + 3 * src->sh_coeffs) * src->Pa * sizeof(float);
    (void)hipDeviceSynchronize();  // a trainer may still be updating the source on its own (non-blocking) stream
    hipError_t e = hipMemcpy(m->planes, src->planes, bytes, hipMemcpyDeviceToDevice);
    if (e != hipSuccess) { gs_model_destroy(m); set_error("clone copy failed: %s", hipGetErrorString(e)); return GS_ERR_HIP; }
    *out = m;
    return GS_OK;
}

extern "C" int gs_model_download(const gs_model* m, float* loc, float* sh, float* scale, float* opac, float* rot) {
    if (!m) return GS_ERR_INVALID_ARGUMENT;
    const int count = m->count, M = m->sh_coeffs;
    if (count == 0) return GS_OK;
    if (!loc || !sh || !scale || !opac || !rot) { set_error("gs_model_download: null destination"); return GS_ERR_INVALID_ARGUMENT; }
    const size_t nf = (size_t)count * (11 + 3 * M);
    GS_HIP(hipDeviceSynchronize());  // a trainer may still be updating these planes on its own (non-blocking) stream
    float* stage = nullptr;
    GS_HIP(hipMalloc((void**)&stage, nf * sizeof(float)));
    float* dl = stage; float* dsh = dl + 3 * (size_t)count; float* dsc = dsh + 3 * (size_t)M * count;
    float* dop = dsc + 3 * (size_t)count; float* dr = dop + count;
    int rc = launch_soa_to_aos(count, m->Pa, M, m->planes, dl, dsh, dsc, dop, dr, 0);
    auto cp = [&](float* h, const float* d, size_t n) { if (rc == GS_OK && hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost) != hipSuccess) rc = GS_ERR_HIP; };
    cp(loc, dl, 3 * (size_t)count); cp(sh, dsh, 3 * (size_t)M * count); cp(scale, dsc, 3 * (size_t)count);
    cp(opac, dop, count); cp(rot, dr, 4 * (size_t)count);
    (void)hipFree(stage);
    if (rc == GS_ERR_HIP) set_error("model download failed: %s", hipGetErrorString(hipGetLastError()));
    return rc;
}

extern "C" int gs_model_info(const gs_model* m, int* capacity, int* sh_degree, int* sh_coeffs, int* count) {
    if (!m) return GS_ERR_INVALID_ARGUMENT;
    if (capacity) *capacity = m->capacity;
    if (sh_degree) *sh_degree = m->sh_degree;
    if (sh_coeffs) *sh_coeffs = m->sh_coeffs;
    if (count) *count = m->count;
    return GS_OK;
}

extern "C" int gs_model_destroy(gs_model* m) {
    if (!m) return GS_OK;
    if (m->planes) (void)hipFree(m->planes);
    delete m;
    return GS_OK;
}

// =============================================================================================
// trainer
// =============================================================================================
struct gs_trainer {
    Options opt;                  // copied from the process defaults at creation; gs_trainer_set_option edits them
    int steps_on_these_lists = 0;  // accumulate calls since the model / views / arena last changed: the longest-list hint below is
                                  // what the device wrote two calls ago and the early flag copy of the last call brought back
    DevBuf densify_work;          // classification flags, ranks and scan partials of densify / prune
    DevBuf spare_planes, spare_m, spare_v;  // the plane sets the last densify replaced: the next one writes into them (no hipMalloc /
                                            // hipFree of three plane sets per densify step: they were a third of its 3.5 ms)
    DevBuf sh16;                  // [3M][Pa] half: read copy of the SH planes (option "sh_fp16")
    const void* sh16_of = nullptr;  // the parameter planes the copy was made from and kept current with (null: stale)
    int device = 0, W = 0, H = 0;
    hipStream_t stream = nullptr;
    gs_model* model = nullptr;
    int V = 0, total_samples = 0;
    std::vector<gs_view> h_views;
    std::vector<char> h_view_block;  // views | group cameras | maps, uploaded each step
    std::vector<int> h_view_group;
    int VG = 0, bwd_pairs = 0, bwd_singles = 0;
    DevBuf truth;                 // [V][N] u32
    DevBuf grad, adam_m, adam_v;  // [(12+3M)][Pa], [(11+3M)][Pa] x2
    int grad_Pa = 0, grad_M = 0, adam_t = 0;
    bool adam_valid = false;
    ScratchSet train, preview;
    uint32_t* h_flags = nullptr;  // pinned, [V][4] + loss[V]
    size_t h_flags_cap = 0;
    hipEvent_t ev_flags = nullptr;   // recorded behind the early copy of the overflow flags
    uint32_t* h_cut = nullptr;       // pinned, [G][4]: the flags again, copied behind the forward of a step whose lists were cut
    size_t h_cut_cap = 0;
    hipEvent_t ev_cut = nullptr;
    bool zcut_tracked = false;       // the newest forward of this configuration left the tiles' depth bounds
    int cut_holdoff = 0, cut_holdoff_next = 2;   // steps without a cut after a wrong one (doubling up to 64, halving again with every good cut)
    long long cut_steps = 0, cut_replays = 0;    // gs_trainer_list_cut_stats
    const void* views_on_device = nullptr;  // where the current view block was last uploaded (null: must upload)
    bool stats_stale = false;        // `last` lacks the device-side numbers (loss, list lengths) of the newest step
    uint32_t Rcap = 0;
    int row_epoch = 0;               // mark of the newest launch's gradient rows (1 .. 255; Scratch::row_epoch), advanced per accumulate attempt
    gs_allreduce_fn allreduce = nullptr;
    void* allreduce_user = nullptr;
    gs_collective_fn shard_rs = nullptr, shard_ag = nullptr;  // sharded update (gs_trainer_set_sharded_update)
    void* shard_user = nullptr;
    int shard_rank = 0, shard_world = 1;
    // compact exchange (gs_trainer_set_compact_exchange): all-gather of dL_dRGB records + all-reduce of the twelve non-SH planes
    gs_collective_fn xchg_gather = nullptr;
    gs_allreduce_fn xchg_reduce = nullptr;
    void* xchg_user = nullptr;
    int xchg_rank = 0, xchg_world = 1, xchg_cameras = 0;
    void* xchg_comms[2] = { nullptr, nullptr };  // gs_trainer_attach_comm_compact: the communicators of the all-gather and of the all-reduce
    DevBuf xgeo, xrgb;
    hipStream_t stream2 = nullptr;           // carries the all-reduce beside the all-gather
    hipEvent_t ev_packed = nullptr, ev_reduced = nullptr;
    gs_step_stats last{};
    bool accumulated = false;
    // optional per-stage HIP-event timing (bench / roofline evidence)
    bool profiling = false;
    struct Pending { int stage; hipEvent_t a, b; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> event_pool, marks;
    hipEvent_t prev_mark = nullptr, stage_start = nullptr;
    int prev_mark_stage = -1;
    unsigned profiling_mask = 0;  // bit s: stage s is timed
    bool roctx_open = false;      // a roctx range of this trainer is open (option "roctx")
    double stage_ms[GS_STAGE_COUNT] = { 0 };
    long long stage_launches[GS_STAGE_COUNT] = { 0 };
};

namespace {
const char* const kStageNames[GS_STAGE_COUNT] = { "preprocess", "scan", "scatter", "tile_sort", "render_forward",
                                                  "render_backward", "splat_backward", "update", "collective" };
// roctx ranges ("roctx ranges name each stage"): the marker library is resolved on first use, so a
// process that never asks for ranges never loads it.  A range brackets the host-side enqueue of the stage's launches, which
// is how rocprofv3 --marker-trace attributes the kernels launched inside it.
struct Roctx { bool tried = false; int (*push)(const char*) = nullptr; int (*pop)() = nullptr; } g_roctx;
bool roctx_ready() {
    if (!g_roctx.tried) {
        g_roctx.tried = true;
        void* lib = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (lib) {
            g_roctx.push = (int (*)(const char*))dlsym(lib, "roctxRangePushA");
            g_roctx.pop = (int (*)())dlsym(lib, "roctxRangePop");
        }
    }
    return g_roctx.push && g_roctx.pop;
}
hipEvent_t prof_event(gs_trainer* t) {
    hipEvent_t e = nullptr;
    if (!t->event_pool.empty()) { e = t->event_pool.back(); t->event_pool.pop_back(); }
    else if (hipEventCreate(&e) != hipSuccess) e = nullptr;
    return e;
}
// Per-stage timing on the trainer's stream.  An event costs ~3 us of stream time on MI355X, so a stage's start
// reuses the end event of the stage that ran directly before it when that one was timed too, and stages outside
// the profiling mask record nothing at all.
void prof_stage_begin(gs_trainer* t, int stage, int stage_before) {
    if (t->opt.roctx && roctx_ready()) { g_roctx.push(kStageNames[stage]); t->roctx_open = true; }
    if (!(t->profiling_mask >> stage & 1u)) return;
    if (t->prev_mark && t->prev_mark_stage == stage_before) { t->stage_start = t->prev_mark; return; }
    hipEvent_t e = prof_event(t);
    t->stage_start = e;
    if (!e) return;
    (void)hipEventRecord(e, t->stream);
    t->marks.push_back(e);
}
void prof_stage_end(gs_trainer* t, int stage) {
    if (t->roctx_open) { g_roctx.pop(); t->roctx_open = false; }
    if (!(t->profiling_mask >> stage & 1u) || !t->stage_start) return;
    hipEvent_t e = prof_event(t);
    if (!e) { t->prev_mark = nullptr; return; }
    (void)hipEventRecord(e, t->stream);
    t->marks.push_back(e);
    t->pending.push_back({ stage, t->stage_start, e });
    t->prev_mark = e; t->prev_mark_stage = stage;
    t->stage_start = nullptr;
}
// "debug_sync": the reference passes debug=true to every rasterizer call (src/Trainer.cu:201,360,412), which makes it
// synchronise and check for errors after each internal kernel; this names the stage a fault belongs to.
int debug_check(gs_trainer* t, int stage) {
    if (!t->opt.debug_sync) return GS_OK;
    hipError_t e = hipStreamSynchronize(t->stream);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) { set_error("stage '%s' failed: %s", kStageNames[stage], hipGetErrorString(e)); return GS_ERR_HIP; }
    return GS_OK;
}
void prof_resolve(gs_trainer* t) {  // caller has synchronised the stream
    for (auto& p : t->pending) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { t->stage_ms[p.stage] += ms; t->stage_launches[p.stage]++; }
    }
    t->pending.clear();
    for (hipEvent_t e : t->marks) t->event_pool.push_back(e);
    t->marks.clear();
    t->prev_mark = nullptr; t->stage_start = nullptr;
}
}  // namespace

extern "C" int gs_trainer_create(int width, int height, gs_trainer** out) {
    if (!out || width <= 0 || height <= 0 || width > 65535 * TILE || height > 65535 * TILE) {
        set_error("gs_trainer_create: bad resolution %dx%d", width, height);
        return GS_ERR_INVALID_ARGUMENT;
    }
    {
        const Dims dd = make_dims(0, 64, 0, 1, width, height, 1, 1, 1.0f);
        if (dd.NST > MAX_SUPER_TILES) {
            set_error("gs_trainer_create: %dx%d has %d super-tiles; this build bins up to %d (8192x8192, 16384x4096, ...)", width, height, dd.NST, MAX_SUPER_TILES);
            return GS_ERR_INVALID_ARGUMENT;
        }
    }
    GS_TRY(require_device());
    gs_trainer* t = new gs_trainer();
    t->opt = g_defaults;
    if (const char* e = getenv("GS_ROCTX")) t->opt.roctx = atoi(e) != 0;
    t->W = width; t->H = height;
    if (hipGetDevice(&t->device) != hipSuccess || hipStreamCreateWithFlags(&t->stream, hipStreamNonBlocking) != hipSuccess) {
        delete t; set_error("stream creation failed: %s", hipGetErrorString(hipGetLastError())); return GS_ERR_HIP;
    }
    // placeholder model so the trainer is never model-less (src/Trainer.cu:111-112)
    int rc = gs_model_create(0, 0, 1, 0, nullptr, nullptr, nullptr, nullptr, nullptr, &t->model);
    if (rc != GS_OK) { (void)hipStreamDestroy(t->stream); delete t; return rc; }
    *out = t;
    return GS_OK;
}

extern "C" int gs_trainer_destroy(gs_trainer* t) {
    if (!t) return GS_OK;
    (void)hipStreamSynchronize(t->stream);
    gs_model_destroy(t->model);
    t->truth.release(); t->grad.release(); t->adam_m.release(); t->adam_v.release(); t->sh16.release(); t->densify_work.release();
    t->spare_planes.release(); t->spare_m.release(); t->spare_v.release();
    t->xgeo.release(); t->xrgb.release();
    if (t->stream2) { (void)hipStreamSynchronize(t->stream2); (void)hipStreamDestroy(t->stream2); }
    if (t->ev_packed) (void)hipEventDestroy(t->ev_packed);
    if (t->ev_reduced) (void)hipEventDestroy(t->ev_reduced);
    t->train.release(); t->preview.release();
    if (t->h_flags) (void)hipHostFree(t->h_flags);
    if (t->ev_flags) (void)hipEventDestroy(t->ev_flags);
    if (t->h_cut) (void)hipHostFree(t->h_cut);
    if (t->ev_cut) (void)hipEventDestroy(t->ev_cut);
    prof_resolve(t);
    for (hipEvent_t e : t->event_pool) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(t->stream);
    delete t;
    return GS_OK;
}

extern "C" int gs_trainer_set_model(gs_trainer* t, gs_model* m) {
    if (!t || !m) return GS_ERR_INVALID_ARGUMENT;
    if (m->device != t->device) { set_error("model lives on device %d, trainer on %d", m->device, t->device); return GS_ERR_INVALID_ARGUMENT; }
    GS_HIP(hipStreamSynchronize(t->stream));
    if (t->model != m) gs_model_destroy(t->model);
    t->model = m;
    t->adam_valid = false; t->adam_t = 0; t->accumulated = false;
    t->sh16_of = nullptr; t->steps_on_these_lists = 0;
    // The densify pool (three spare plane sets, each sized for the model it last served + 50 %: ~1 GB at 1M splats, M = 16) is
    // kept only while it fits the model in use: a smaller model gives the memory back (the next densify allocates afresh).
    const size_t need = plane_buffer_floats(m->sh_coeffs, m->Pa) * sizeof(float);
    for (DevBuf* b : { &t->spare_planes, &t->spare_m, &t->spare_v })
        if (b->cap > 2 * need + (64u << 20)) b->release();
    return GS_OK;
}
extern "C" gs_model* gs_trainer_get_model(gs_trainer* t) { return t ? t->model : nullptr; }

extern "C" int gs_trainer_set_views(gs_trainer* t, int n_views, const gs_view* views, const uint32_t* const* truth,
                                    int truth_on_device, int total_samples) {
    if (!t || n_views < 0 || (n_views > 0 && (!views || !truth)) || total_samples < n_views) {
        set_error("gs_trainer_set_views: bad arguments");
        return GS_ERR_INVALID_ARGUMENT;
    }
    GS_HIP(hipStreamSynchronize(t->stream));
    const size_t N = (size_t)t->W * t->H;
    GS_TRY(t->truth.ensure(std::max<size_t>(1, (size_t)n_views * N * 4)));
    for (int v = 0; v < n_views; v++) {
        if (!truth[v]) { set_error("gs_trainer_set_views: truth image %d is null", v); return GS_ERR_INVALID_ARGUMENT; }
        GS_HIP(hipMemcpy(t->truth.as<uint32_t>() + (size_t)v * N, truth[v], N * 4,
                         truth_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    }
    t->h_views.assign(views, views + n_views);
    t->V = n_views;
    t->VG = build_view_block(views, n_views, t->opt.share != 0, t->h_view_block, &t->bwd_pairs, &t->bwd_singles);
    {
        const int* vg = reinterpret_cast<const int*>(t->h_view_block.data() + (size_t)n_views * 2 * sizeof(gs_view));
        t->h_view_group.assign(vg, vg + n_views);
    }
    t->total_samples = total_samples;
    t->accumulated = false;
    t->views_on_device = nullptr;
    t->stats_stale = false;
    t->steps_on_these_lists = 0;
    return GS_OK;
}

static int trainer_dims(gs_trainer* t, Dims* d) {
    int D = 0;
    GS_TRY(effective_degree(t->model->sh_degree, t->model->sh_coeffs, &D));
    *d = make_dims(t->model->count, t->model->Pa, D, t->model->sh_coeffs, t->W, t->H, t->V, t->Rcap, 1.0f, t->VG, t->opt.cull);
    return GS_OK;
}

// The device-side numbers of the newest step (loss, entry counts, longest tile list) are fetched on demand, so a step
// nobody asks about costs no end-of-step synchronisation.
static int resolve_stats(gs_trainer* t) {
    GS_HIP(hipStreamSynchronize(t->stream));
    prof_resolve(t);
    if (!t->stats_stale) return GS_OK;
    const int V = t->V;
    if (t->model && t->model->count > 0) {  // an empty model ran no backward pass: its loss statistic stays zero
        Dims d;
        GS_TRY(trainer_dims(t, &d));
        GS_TRY(launch_loss_sum(d, t->train.s, t->stream));
        GS_HIP(hipStreamSynchronize(t->stream));
    }
    GS_HIP(hipMemcpy(t->h_flags, t->train.s.flags, (size_t)V * 20, hipMemcpyDeviceToHost));
    gs_step_stats& st = t->last;
    st.num_rendered = 0; st.max_tile_list = 0;
    for (int g = 0; g < t->VG; g++) st.max_tile_list = std::max(st.max_tile_list, (int)t->h_flags[g * 4 + 1]);
    for (int v = 0; v < V; v++) st.num_rendered += t->h_flags[t->h_view_group[v] * 4 + 2];  // R of every pass, as the reference counts
    const float* hl = reinterpret_cast<const float*>(t->h_flags + (size_t)V * 4);
    double L = 0;
    if (t->model && t->model->count > 0)
        for (int v = 0; v < V; v++) L += hl[v];
    st.loss = (float)L;
    t->stats_stale = false;
    return GS_OK;
}

// Launches one iteration's projection, binning, forward, loss and backward for all passes and leaves the averaged
// gradients in t->grad.  The only host wait is on the arena-overflow flags, which the device publishes right after the
// projection (a few tens of microseconds into the step, while the rest of the step is already queued behind it): on
// return the stream is still busy.  An overflowing arena is grown and the iteration replayed.
// xchg != null (compact exchange): the per-splat stage leaves the rank's share of the exchange in *xchg's buffers instead of
// gradient planes (k_sh_rebuild writes those after the collectives); xchg->slots is set here for the form the step takes.
// fuse_h != null (gs_trainer_step without a collective): the update is applied by the per-splat reduction of the same attempt (FusedUpdate);
// *update_applied says whether it was (a step with ONE record per splat runs the chain and the reduction in one kernel, which reads
// the parameters it would have to write: that form keeps the update launch).
static int prepare_adam(gs_trainer* t, const gs_hyper* h);
static int accumulate_async(gs_trainer* t, bool need_var, Exchange* xchg = nullptr, const gs_hyper* fuse_h = nullptr, bool* update_applied = nullptr) {
    if (update_applied) *update_applied = false;
    if (t->V == 0 && t->total_samples <= 0) { set_error("Can't run training iteration, no truth data available!"); return GS_ERR_NO_TRUTH; }
    if (!t->model) return GS_ERR_NO_MODEL;
    GS_HIP(hipSetDevice(t->device));
    gs_model* m = t->model;
    const int P = m->count, M = m->sh_coeffs, V = t->V;
    const Planes pl{ M };
    if (V == 0) {
        // A data-parallel rank that owns no pass of this iteration (more ranks than passes): its contribution to the
        // averaged gradients is zero, and it still joins the collective and applies the common update.
        GS_TRY(t->grad.ensure(plane_buffer_floats(M, m->Pa) * 4));
        t->grad_Pa = m->Pa; t->grad_M = M;
        GS_HIP(hipMemsetAsync(t->grad.p, 0, plane_buffer_floats(M, m->Pa) * 4, t->stream));  // the chunk padding behind the planes too: it takes part in the collectives
        gs_step_stats st0{};
        st0.count_before = st0.count_after = P;
        t->last = st0; t->stats_stale = false; t->accumulated = true;
        return GS_OK;
    }
    if (t->pending.size() > 4096) { GS_HIP(hipStreamSynchronize(t->stream)); prof_resolve(t); }
    gs_step_stats st{};
    st.count_before = st.count_after = P; st.views = V;
    // gradient planes (+ var), sized to the model's plane stride
    {
        const uint64_t before = t->grad.generation;      // (not the pointer: a regrown buffer may come back at the old address)
        GS_TRY(t->grad.ensure(plane_buffer_floats(M, m->Pa) * 4));
        if (t->grad.generation != before || t->grad_Pa != m->Pa)   // the tail behind the planes takes part in the collectives' chunks: keep it finite
            GS_HIP(hipMemsetAsync(t->grad.as<float>() + (size_t)(pl.count() + 1) * m->Pa, 0, 64 * 64 * 4, t->stream));
    }
    t->grad_Pa = m->Pa; t->grad_M = M;
    if (t->Rcap == 0) t->Rcap = t->opt.arena ? (uint32_t)t->opt.arena : (uint32_t)std::max<long long>(1 << 20, 16LL * P);
    if (t->h_flags_cap < (size_t)V * 20) {
        GS_HIP(hipStreamSynchronize(t->stream));
        if (t->h_flags) (void)hipHostFree(t->h_flags);
        t->h_flags = nullptr;
        GS_HIP(hipHostMalloc((void**)&t->h_flags, (size_t)V * 20));
        t->h_flags_cap = (size_t)V * 20;
    }
    if (!t->ev_flags) GS_HIP(hipEventCreateWithFlags(&t->ev_flags, hipEventDisableTiming));
    // Depth cut of the tile lists (Dims::cut).  In a dense scene a tile's pixels finish long before its list ends (1M splats @2048^2:
    // 170 of 1100 entries looked at), and from one training step to the next the model barely moves: the forward leaves, per tile, the
    // depth behind which nothing was read, and the NEXT step's binning lists only what lies in front of it (+ a margin) — the scatter
    // and the sort then handle a third of the entries.  The forward checks its own premise (every pixel of a shortened list must
    // finish inside it); if it does not hold the camera's backward, per-splat stage and fused update skip themselves on the device,
    // the host — which waits for that verdict behind the forward of a cut step, the rest of the step already queued — replays the
    // step uncut and holds off for a few steps.  A cut step that stands is the uncut step bit for bit: same blend, same rows, same sums.
    // Used where it pays (previous step: list_cut_min_avg entries per tile and more), with row marks (a dropped entry owns no row), on
    // lists that saw at least one forward in this configuration (model, views, arena unchanged).
    // The forward of a step TRACKS the bounds (k_render_fwd<true>) only where the next step may use them — a scene that is dense by the previous
    // step's entry count — and a step cuts only lists whose bounds the step before it tracked in this configuration; everywhere else (cfg1-cfg4)
    // every kernel is the form without any of this.
    bool cut = false, track = false;
    if (t->opt.list_cut && t->opt.row_marks != 0 && t->steps_on_these_lists >= 1 && P > 0 && t->VG > 0) {
        unsigned long long entries = 0;
        for (int g = 0; g < t->VG; g++) entries += t->h_flags[g * 4 + 2];
        const unsigned long long tiles = (unsigned long long)t->VG * (unsigned long long)(((t->W + TILE - 1) / TILE) * ((t->H + TILE - 1) / TILE));
        track = entries >= (unsigned long long)t->opt.list_cut_min_avg * tiles;
        if (track && t->zcut_tracked) {
            if (t->cut_holdoff > 0) t->cut_holdoff--;
            else cut = true;
        }
    }
    if (t->steps_on_these_lists == 0) t->zcut_tracked = false;
    if (cut) {
        if (t->h_cut_cap < (size_t)t->VG * 16) {
            if (t->h_cut) (void)hipHostFree(t->h_cut);
            t->h_cut = nullptr;
            GS_HIP(hipHostMalloc((void**)&t->h_cut, (size_t)t->VG * 16));
            t->h_cut_cap = (size_t)t->VG * 16;
        }
        if (!t->ev_cut) GS_HIP(hipEventCreateWithFlags(&t->ev_cut, hipEventDisableTiming));
    }
    for (;;) {
        Dims d;
        GS_TRY(trainer_dims(t, &d));
        GS_TRY(t->train.ensure(P, V, t->W, t->H, t->Rcap, true));
        d.Rcap = t->train.Rcap;
        d.cut = cut ? 1 : 0;
        d.cut_track = track ? 1 : 0;
        d.cut_margin = t->opt.list_cut_margin;
        // The long-list sort launch is empty in most scenes (no tile list reaches SORT_SMALL_CAP entries) and still costs a
        // dependent launch — 7 us of a 255 us step at the 8-GPU load.  The host knows the longest list of two steps ago (the
        // device writes it after the early flag copy, which therefore carries the previous step's value): with a quarter of
        // headroom the launch is skipped; a list that outgrows the hint anyway is sorted by k_tile_build_sort's global-scratch
        // path — slower, never wrong.
        // The same holds one class down: k_tile_sort_mid takes the lists of SORT_TINY_CAP entries and more.
        if (t->opt.long_sort >= 0) d.long_sort = d.mid_sort = t->opt.long_sort;
        else if (t->steps_on_these_lists >= 2) {
            uint32_t longest = 0;
            for (int g = 0; g < t->VG; g++) longest = std::max(longest, t->h_flags[g * 4 + 1]);
            d.long_sort = longest >= (uint32_t)(SORT_SMALL_CAP - SORT_SMALL_CAP / 4) ? 1 : 0;
            d.mid_sort = longest >= (uint32_t)(SORT_TINY_CAP - SORT_TINY_CAP / 4) ? 1 : 0;
            // Grids by the same kind of hint (flags[3]: where the lists of SORT_TINY_CAP entries and more end in the tile order, and
            // where the first shorter one sits, two steps ago): the short-list sorter is not launched over a head that holds no short
            // list (a dense scene: 65k workgroups that would look at a tile and leave, 45 us), k_tile_sort_mid gets a workgroup
            // per tile of its head and no more.  Stale hints cost speed only: k_tile_sort_mid walks on in strides past its grid
            // and sorts the short lists in front of small_first itself.
            if (d.mid_sort) {
                const uint32_t unit = order_hint_unit(d.T);
                uint32_t mid_end = 0, small_start = 0xFFFFFFFFu;
                for (int g = 0; g < t->VG; g++) {
                    mid_end = std::max(mid_end, (t->h_flags[g * 4 + 3] >> 16) * unit);
                    small_start = std::min(small_start, (t->h_flags[g * 4 + 3] & 0xFFFFu) * unit);
                }
                const uint32_t keep = std::max(64u, small_start / 8);
                d.small_first = small_start > keep ? (int)std::min((uint32_t)d.T, small_start - keep) : 0;
                d.mid_grid = (int)std::min((uint32_t)d.T, mid_end + std::max(64u, mid_end / 8));
            }
        }
        if (t->opt.sort_grids >= 0) {  // test hook: any grids must give the same lists (tests/test_gpu_trainer.py)
            d.mid_sort = 1;
            d.small_first = std::min(d.T, t->opt.sort_grids & 0xFFFF);
            d.mid_grid = std::max(1, std::min(d.T, t->opt.sort_grids >> 16));
        }
        // Row marks: in a scene with long tile lists most entries lie behind their tile's last contributor and own an all-zero
        // gradient row — with marks such rows are neither written nor read (cfg5: 84 % of them).  Where lists are short nearly
        // every row exists and the marks only cost (cfg3: +20 us in k_splat_bwd_view).  The kernels decide per camera from the
        // longest tile list of THIS step (the tile scan writes it before the backward runs; uses_row_marks): marks from 1024
        // entries on.  Either way the gradients are the same bits.  The host only supplies a fresh epoch per attempt and clears
        // all marks when the epochs wrap or the buffer is new.
        d.epoch = 0; d.marks_min_list = (t->opt.row_marks < 0 && !d.cut) ? 1024u : 0u;   // (an entry the cut dropped owns no gradient row: marks always)
        if (t->opt.row_marks != 0) {
            if (t->row_epoch >= 255 || t->train.rowmark_cleared != t->train.rowmark.generation) {
                GS_HIP(hipMemsetAsync(t->train.rowmark.p, 0, t->train.rowmark.cap, t->stream));
                t->train.rowmark_cleared = t->train.rowmark.generation;
                t->row_epoch = 0;
            }
            d.epoch = ++t->row_epoch;
        }
        Scratch s = t->train.s;
        s.truth = t->truth.as<uint32_t>();
        if (!t->opt.reuse_masks) s.hit_masks = nullptr;  // the forward stores no ballots, the backward tests the blocks itself
        if (t->opt.sh_fp16 && P > 0) {
            // the half-precision read copy: made here when it does not belong to these planes (new model, densify, a
            // sharded update that refreshed only this rank's chunk), otherwise kept current by the update kernel
            if (t->sh16_of != (const void*)m->planes) {
                GS_TRY(t->sh16.ensure((size_t)3 * M * m->Pa * sizeof(uint16_t)));
                GS_TRY(launch_sh_to_half(M, P, m->Pa, m->planes, t->sh16.as<uint16_t>(), t->stream));
                t->sh16_of = (const void*)m->planes;
            }
            s.sh16 = t->sh16.as<uint16_t>();
        }
        if (t->views_on_device != (const void*)s.views) {  // the view block only changes with gs_trainer_set_views
            GS_HIP(hipMemcpyAsync((void*)s.views, t->h_view_block.data(), t->h_view_block.size(), hipMemcpyHostToDevice, t->stream));
            t->views_on_device = (const void*)s.views;
        }
        prof_stage_begin(t, 0, -1);
        if (d.cut) GS_TRY(launch_stile_zcut(d, s, t->stream));
        GS_TRY(launch_preprocess(d, m->planes, s, t->stream));
        prof_stage_end(t, 0);
        GS_TRY(debug_check(t, 0));
        prof_stage_begin(t, 1, 0);
        GS_TRY(launch_coarse_colscan(d, s, t->stream));
        GS_HIP(hipMemcpyAsync(t->h_flags, s.flags, (size_t)t->VG * 16, hipMemcpyDeviceToHost, t->stream));
        GS_HIP(hipEventRecord(t->ev_flags, t->stream));
        prof_stage_end(t, 1);  // the scans and the publication of their overflow verdict
        GS_TRY(debug_check(t, 1));
        prof_stage_begin(t, 2, 1);
        GS_TRY(stage_bin(d, s, t->stream));
        prof_stage_end(t, 2);
        GS_TRY(debug_check(t, 2));
        prof_stage_begin(t, 3, 2);
        GS_TRY(launch_tile_build_sort(d, s, t->stream));
        prof_stage_end(t, 3);
        GS_TRY(debug_check(t, 3));
        prof_stage_begin(t, 4, 3);
        GS_TRY(launch_render_forward(d, s, t->stream));
        if (d.cut) {   // the forward's verdict on the cut, for the host (below)
            GS_HIP(hipMemcpyAsync(t->h_cut, s.flags, (size_t)t->VG * 16, hipMemcpyDeviceToHost, t->stream));
            GS_HIP(hipEventRecord(t->ev_cut, t->stream));
        }
        prof_stage_end(t, 4);
        GS_TRY(debug_check(t, 4));
        if (P > 0) {  // an empty model has no gradients: its image is the background, its loss the residual against it
            const int* items = reinterpret_cast<const int*>(reinterpret_cast<const char*>(s.views) + view_block_items_offset(V));
            prof_stage_begin(t, 5, 4);
            const bool fuse = !need_var && t->opt.fuse != 0;
            GS_TRY(launch_render_backward(d, s, items, t->bwd_pairs, t->bwd_singles, fuse, t->stream));
            prof_stage_end(t, 5);
            GS_TRY(debug_check(t, 5));
            prof_stage_begin(t, 6, 5);
            if (xchg) xchg->slots = ((t->xchg_cameras + t->xchg_world - 1) / t->xchg_world) * (fuse ? 1 : 2);
            FusedUpdate fu;
            if (fuse_h) {
                GS_TRY(prepare_adam(t, fuse_h));      // moments allocated and zeroed on the first Adam step; the step counter moves after the attempt
                fu.u = make_update_args(pl, *fuse_h, t->adam_t + 1);
                fu.params = m->planes; fu.am = t->adam_m.as<float>(); fu.av = t->adam_v.as<float>();
                fu.sh16 = (t->opt.sh_fp16 && t->sh16_of == (const void*)m->planes) ? t->sh16.as<uint16_t>() : nullptr;
            }
            bool applied = false;
            GS_TRY(launch_splat_backward_avg(d, m->planes, s, (float)t->total_samples, t->grad.as<float>(), items, t->bwd_pairs,
                                             t->bwd_singles, fuse, t->stream, xchg, fuse_h ? &fu : nullptr, &applied));
            if (update_applied) *update_applied = applied;
            prof_stage_end(t, 6);
            GS_TRY(debug_check(t, 6));
        }
        GS_HIP(hipEventSynchronize(t->ev_flags));
        bool overflow = false;
        uint32_t need = 0;
        for (int g = 0; g < t->VG; g++) {
            if (t->h_flags[g * 4 + 0] & 1u) overflow = true;
            need = std::max(need, t->h_flags[g * 4 + 2]);
        }
        if (!overflow && d.cut) {
            GS_HIP(hipEventSynchronize(t->ev_cut));
            bool wrong = false;
            for (int g = 0; g < t->VG; g++) wrong = wrong || (t->h_cut[g * 4 + 0] & 2u);
            t->cut_steps++;
            if (wrong) {   // replay uncut (nothing of this attempt was applied); no cut for a while
                t->cut_replays++;
                t->cut_holdoff = t->cut_holdoff_next;
                t->cut_holdoff_next = std::min(64, t->cut_holdoff_next * 2);
                cut = false;
                continue;
            }
            t->cut_holdoff_next = std::max(2, t->cut_holdoff_next / 2);
        }
        if (!overflow) break;
        GS_HIP(hipStreamSynchronize(t->stream));  // the overflowed groups' later stages are no-ops; drain them before regrowing
        prof_resolve(t);
        t->Rcap = (uint32_t)std::min<unsigned long long>(0xFFFFFF00ull, (unsigned long long)need + need / 4 + 1024);
        t->steps_on_these_lists = 0;
        st.arena_regrows++;
        if (st.arena_regrows > 8) { set_error("binning arena failed to converge"); return GS_ERR_INTERNAL; }
    }
    t->last = st;
    t->stats_stale = true;
    t->accumulated = true;
    t->steps_on_these_lists++;
    t->zcut_tracked = track;      // the attempt that stands wrote every tile's bound (or none did)
    if (update_applied && *update_applied) {   // the attempt that went through applied the update (an overflowed one applies nothing: k_splat_bwd_reduce)
        if (fuse_h->update_rule == GS_UPDATE_ADAM) t->adam_t++;
        if (!(t->opt.sh_fp16 && t->sh16_of == (const void*)m->planes)) t->sh16_of = nullptr;
        t->accumulated = false;
    }
    return GS_OK;
}

extern "C" int gs_trainer_accumulate(gs_trainer* t, gs_step_stats* stats) {
    if (!t) return GS_ERR_INVALID_ARGUMENT;
    GS_TRY(accumulate_async(t, true));
    GS_TRY(resolve_stats(t));  // callers of the split API read the gradient buffer next: hand it over complete
    if (stats) *stats = t->last;
    return GS_OK;
}

extern "C" int gs_trainer_list_cut_stats(gs_trainer* t, long long* steps_cut, long long* replays) {
    if (!t) return GS_ERR_INVALID_ARGUMENT;
    if (steps_cut) *steps_cut = t->cut_steps;
    if (replays) *replays = t->cut_replays;
    return GS_OK;
}
// Diagnostic (synchronises): what the newest step's binning actually listed — out[0] candidates (splat, super-tile) emitted by the coarse
// scatter, out[1] tile-list entries, out[2] tiles with a finite depth bound for the NEXT step, out[3] tiles — summed over the cameras.
extern "C" int gs_trainer_debug_list_totals(gs_trainer* t, long long out[4]) {
    if (!t || !out) return GS_ERR_INVALID_ARGUMENT;
    GS_HIP(hipSetDevice(t->device));
    GS_HIP(hipStreamSynchronize(t->stream));
    const ScratchSet& ss = t->train;
    const size_t G = (size_t)std::max(t->VG, 0);
    std::vector<uint32_t> cc(G * ss.NST), tc(G * ss.T), zc(G * ss.T);
    if (G && ss.s.coarse_count) GS_HIP(hipMemcpy(cc.data(), ss.s.coarse_count, cc.size() * 4, hipMemcpyDeviceToHost));
    if (G && ss.s.tile_count) GS_HIP(hipMemcpy(tc.data(), ss.s.tile_count, tc.size() * 4, hipMemcpyDeviceToHost));
    if (G && ss.s.tile_zcut) GS_HIP(hipMemcpy(zc.data(), ss.s.tile_zcut, zc.size() * 4, hipMemcpyDeviceToHost));
    out[0] = out[1] = out[2] = 0; out[3] = (long long)tc.size();
    for (uint32_t x : cc) out[0] += x;
    for (uint32_t x : tc) out[1] += x;
    if (ss.s.tile_zcut) for (uint32_t x : zc) out[2] += x != 0xFFFFFFFFu;
    return GS_OK;
}

extern "C" int gs_trainer_grad_buffer(gs_trainer* t, float** p, size_t* n) {
    if (!t || !t->model) return GS_ERR_INVALID_ARGUMENT;
    const Planes pl{ t->model->sh_coeffs };
    GS_TRY(t->grad.ensure(plane_buffer_floats(t->model->sh_coeffs, t->model->Pa) * 4));
    if (p) *p = t->grad.as<float>();
    if (n) *n = (size_t)(pl.count() + 1) * t->model->Pa;
    return GS_OK;
}

// Densify / prune on the device (k_densify.hip): classify + three prefix sums, one read-back of the three totals to
// size the new planes, one emit pass.  The model object keeps its identity (callers hold the pointer).
static int trainer_densify(gs_trainer* t, const gs_hyper* h, gs_step_stats* st) {
    gs_model* m = t->model;
    const int M = m->sh_coeffs, cap = m->capacity, count = m->count;
    st->count_after = count;
    if (count == 0) return GS_OK;
    const int fs = round_up(count, 64);
    DevBuf& work = t->densify_work;  // flags[3][fs] | ranks[3][fs] | scan partials (kept between densify steps, grow-only)
    const size_t npart = scan_partials_count(count, 3) + 64;
    GS_TRY(work.ensure(((size_t)6 * fs + npart) * 4));
    uint32_t* flags = work.as<uint32_t>();
    uint32_t* ranks = flags + (size_t)3 * fs;
    uint32_t* partials = ranks + (size_t)3 * fs;
    int rc = launch_densify_classify(count, m->Pa, M, m->planes, t->grad.as<float>(), *h, flags, ranks, fs, partials, t->stream);
    uint32_t totals[3] = { 0, 0, 0 };
    for (int k = 0; k < 3 && rc == GS_OK; k++)
        if (hipMemcpyAsync(&totals[k], ranks + (size_t)k * fs + (count - 1), 4, hipMemcpyDeviceToHost, t->stream) != hipSuccess) rc = GS_ERR_HIP;
    if (rc == GS_OK && hipStreamSynchronize(t->stream) != hipSuccess) rc = GS_ERR_HIP;
    if (rc != GS_OK) { if (rc == GS_ERR_HIP) set_error("densify failed: %s", hipGetErrorString(hipGetLastError())); return rc; }
    const int n_split = (int)totals[0], n_clone = (int)totals[1], kept = (int)totals[2];
    const int splits_done = std::min(n_split, std::max(0, cap - count));
    const int clones_done = std::min(n_clone, std::max(0, cap - count - splits_done));
    const int n2 = kept + splits_done + clones_done;
    // The new plane sets (parameters, and the two Adam moments) go into the trainer's spare buffers — the sets the previous densify
    // replaced — and the sets replaced now become the spares: a buffer is only allocated when the model has outgrown the spare,
    // then with half as much again as headroom (never more than the model's capacity needs).
    const int Pa2 = std::max(64, round_up(n2, 64));
    const size_t bytes = plane_buffer_floats(M, Pa2) * sizeof(float);
    const size_t bytes_cap = plane_buffer_floats(M, std::max(64, round_up(std::max(cap, n2), 64))) * sizeof(float);
    auto spare = [&](DevBuf& b) { return b.cap >= bytes ? GS_OK : b.ensure(std::min(bytes_cap, bytes + bytes / 2)); };
    rc = spare(t->spare_planes);
    if (rc == GS_OK && hipMemsetAsync(t->spare_planes.p, 0, bytes, t->stream) != hipSuccess) rc = GS_ERR_HIP;  // in order with the emit pass
    if (rc == GS_OK) rc = launch_densify_emit(count, m->Pa, M, m->planes, t->grad.as<float>(), *h, flags, ranks, fs, splits_done, clones_done, kept,
                                              Pa2, t->spare_planes.as<float>(), t->stream);
    // Adam moments follow their splats (twins inherit the parent's); the step counter keeps running
    if (rc == GS_OK && t->adam_valid) {
        rc = spare(t->spare_m);
        if (rc == GS_OK) rc = spare(t->spare_v);
        if (rc == GS_OK && (hipMemsetAsync(t->spare_m.p, 0, bytes, t->stream) != hipSuccess || hipMemsetAsync(t->spare_v.p, 0, bytes, t->stream) != hipSuccess)) rc = GS_ERR_HIP;
        if (rc == GS_OK) rc = launch_densify_carry(count, m->Pa, M, *h, flags, ranks, fs, splits_done, clones_done, kept, Pa2,
                                                   t->adam_m.as<float>(), t->spare_m.as<float>(), t->adam_v.as<float>(), t->spare_v.as<float>(), t->stream);
    }
    if (rc != GS_OK) {
        if (rc == GS_ERR_HIP) set_error("densify failed: %s", hipGetErrorString(hipGetLastError()));
        (void)hipStreamSynchronize(t->stream);  // nothing was swapped: the model and the moments are as before
        return rc;
    }
    // swap: the model keeps its identity, the replaced sets wait for the next densify.  Everything is ordered on the trainer's
    // stream, so no wait is needed for the sets that were read; a model shared with other streams is the caller's to synchronise,
    // as before a gs_model_destroy.
    DevBuf old_planes;
    old_planes.p = m->planes; old_planes.cap = m->planes_bytes;
    m->planes = t->spare_planes.as<float>(); m->planes_bytes = t->spare_planes.cap;
    t->spare_planes = old_planes;  // DevBuf is a plain (pointer, capacity) pair: ownership moves
    m->Pa = Pa2; m->count = n2;
    t->sh16_of = nullptr; t->steps_on_these_lists = 0;
    if (t->adam_valid) { std::swap(t->adam_m, t->spare_m); std::swap(t->adam_v, t->spare_v); }
    st->count_after = n2;
    return GS_OK;
}

// The parameter update on the flat element range [lo, hi) of the planes (everything by default): Adam state is created
// on first use and its step counter advanced once per call.
// Adam moments exist from the first Adam step on (zeroed, step counter 0); any other rule but the reference's is refused.
static int prepare_adam(gs_trainer* t, const gs_hyper* h) {
    gs_model* m = t->model;
    if (h->update_rule == GS_UPDATE_ADAM) {
        const size_t bytes = plane_buffer_floats(m->sh_coeffs, m->Pa) * 4;
        if (!t->adam_valid) {
            GS_TRY(t->adam_m.ensure(bytes)); GS_TRY(t->adam_v.ensure(bytes));
            GS_HIP(hipMemsetAsync(t->adam_m.p, 0, bytes, t->stream));
            GS_HIP(hipMemsetAsync(t->adam_v.p, 0, bytes, t->stream));
            t->adam_valid = true; t->adam_t = 0;
        }
    } else if (h->update_rule != GS_UPDATE_SGD_CLAMP) {
        set_error("unknown update rule %d", h->update_rule);
        return GS_ERR_INVALID_ARGUMENT;
    }
    return GS_OK;
}
static int apply_update(gs_trainer* t, const gs_hyper* h, int stage_before, size_t lo = 0, size_t hi = ~(size_t)0) {
    gs_model* m = t->model;
    const Planes pl{ m->sh_coeffs };
    GS_TRY(prepare_adam(t, h));
    if (h->update_rule == GS_UPDATE_ADAM) t->adam_t++;
    prof_stage_begin(t, 7, stage_before);
    // the update keeps the fp16 SH copy current for the elements it touches; a partial (sharded) update leaves the other
    // ranks' chunks to the all-gather, so the copy is rebuilt before the next projection
    const bool whole = lo == 0 && hi >= (size_t)pl.count() * m->Pa;
    uint16_t* sh16 = (t->opt.sh_fp16 && whole && t->sh16_of == (const void*)m->planes) ? t->sh16.as<uint16_t>() : nullptr;
    if (!sh16) t->sh16_of = nullptr;
    GS_TRY(launch_update(pl, m->count, m->Pa, m->planes, t->grad.as<float>(), t->adam_m.as<float>(), t->adam_v.as<float>(),
                         t->adam_t, *h, t->stream, lo, hi, sh16));
    prof_stage_end(t, 7);
    GS_TRY(debug_check(t, 7));
    t->accumulated = false;
    return GS_OK;
}

extern "C" int gs_trainer_apply(gs_trainer* t, const gs_hyper* h, int densify, gs_step_stats* stats) {
    if (!t || !h) return GS_ERR_INVALID_ARGUMENT;
    if (!t->accumulated) { set_error("gs_trainer_apply called without gs_trainer_accumulate"); return GS_ERR_INVALID_ARGUMENT; }
    GS_HIP(hipSetDevice(t->device));
    GS_TRY(apply_update(t, h, t->allreduce ? 8 : 6));
    if (densify) {
        GS_TRY(resolve_stats(t));
        GS_TRY(trainer_densify(t, h, &t->last));
    }
    if (stats) { GS_TRY(resolve_stats(t)); *stats = t->last; }
    return GS_OK;
}

// One collective of the sharded update, timed as part of the "collective" stage (stage_before: the stage that ran last).
// A hook that fails has failed on THIS rank only: the error goes back to the caller, who must end the process (bench.py
// exits non-zero, which makes the launcher end the peers) — the library's own communicator is aborted by its hooks.
static int shard_call(gs_trainer* t, gs_collective_fn fn, float* buf, size_t n, const char* what, int stage_before) {
    prof_stage_begin(t, 8, stage_before);
    const int rc = fn(buf, n, (void*)t->stream, t->shard_user);
    prof_stage_end(t, 8);
    if (rc != 0) { set_error("%s hook failed with %d", what, rc); return GS_ERR_COLLECTIVE; }
    return GS_OK;
}

// The data-parallel step with the compact exchange.  48 of cfg3's 60 gradient planes are SH gradients, and a record's SH
// gradient of a splat is rank one (basis(view direction)[M] x dL_dRGB[3], src/Trainer.cu:51-76 as k_splat_bwd_reduce rebuilds
// it): instead of all-reducing (12 + 3M) P floats the ranks all-gather their records' dL_dRGB (3 floats per record and splat)
// and all-reduce the twelve other planes; every rank then rebuilds the SH planes from ALL records in the single-GPU order.
//   main stream:    ... per-splat backward + pack -> [all-gather dL_dRGB] -> k_sh_rebuild (SH planes) -> wait -> (other planes) -> update
//   second stream:                     wait(packed) -> [all-reduce of the twelve other planes] ---------> (reduced)
// cfg3 on 8 ranks: 8.4 + 8.4 MB received per rank instead of the 42 MB a ring all-reduce of 24 MB moves; the SH gradients
// are the single-GPU step's bit for bit (the collective sums only the geometry planes).
static int step_compact_exchange(gs_trainer* t, const gs_hyper* h, int densify, gs_step_stats* stats) {
    if (!t->model) return GS_ERR_NO_MODEL;
    GS_HIP(hipSetDevice(t->device));
    gs_model* m = t->model;
    const int C_ = t->xchg_cameras, G = t->xchg_world;
    const int mine = (C_ - t->xchg_rank + G - 1) / G;   // cameras c with c % world == rank
    if (t->bwd_singles != 0 || t->VG != mine || t->V != 2 * mine || t->total_samples != 2 * C_) {
        set_error("compact exchange: rank %d of %d must hold both passes of its %d of the iteration's %d cameras (camera c on rank c %% world) "
                  "and total_samples = 2 x cameras; the views set are %d passes in %d camera groups (%d unpaired), total_samples %d",
                  t->xchg_rank, G, mine, C_, t->V, t->VG, t->bwd_singles, t->total_samples);
        return GS_ERR_INVALID_ARGUMENT;
    }
    const int cmax = (C_ + G - 1) / G;
    const size_t Pa = (size_t)m->Pa;
    GS_TRY(t->xgeo.ensure(12 * Pa * 4));
    const int hdr = exchange_header_floats(C_, G);
    GS_TRY(t->xrgb.ensure((size_t)G * ((size_t)hdr + 2 * cmax * 3 * Pa) * 4));
    Exchange x;
    x.geo = t->xgeo.as<float>(); x.rgb = t->xrgb.as<float>(); x.rank = t->xchg_rank; x.world = G; x.hdr = hdr;
    GS_TRY(accumulate_async(t, densify != 0, &x));   // sets x.slots for the form this step takes
    const bool per_pass = x.slots == 2 * cmax;
    bool update_done = false;
    if (m->count > 0) {
        if (!t->stream2) GS_HIP(hipStreamCreateWithFlags(&t->stream2, hipStreamNonBlocking));
        if (!t->ev_packed) { GS_HIP(hipEventCreateWithFlags(&t->ev_packed, hipEventDisableTiming)); GS_HIP(hipEventCreateWithFlags(&t->ev_reduced, hipEventDisableTiming)); }
        const bool overlap = t->opt.xchg_overlap != 0;
        // Every early return below leaves the stage / roctx range closed, and — once the all-reduce is under way on the second stream —
        // waits for it: the next step's pack kernel writes x.geo on the main stream and must not meet a collective still reading it.
        struct StageScope {
            gs_trainer* t; int stage; bool open = true;
            StageScope(gs_trainer* t_, int stage_, int before) : t(t_), stage(stage_) { prof_stage_begin(t, stage, before); }
            void end() { if (open) { prof_stage_end(t, stage); open = false; } }
            ~StageScope() { end(); }
        };
        struct SecondStreamGuard {
            gs_trainer* t; bool armed = false;
            ~SecondStreamGuard() { if (armed && t->stream2) (void)hipStreamSynchronize(t->stream2); }
        } second{ t };
        {
            StageScope stage(t, 8, 6);
            int rc = 0;
            if (overlap) {
                GS_HIP(hipEventRecord(t->ev_packed, t->stream));
                GS_HIP(hipStreamWaitEvent(t->stream2, t->ev_packed, 0));
                second.armed = true;
                rc = t->xchg_reduce(x.geo, 12 * Pa, (void*)t->stream2, t->xchg_user);
                if (rc == 0) GS_HIP(hipEventRecord(t->ev_reduced, t->stream2));
            }
            const char* failed = rc != 0 ? "all-reduce (geometry planes)" : nullptr;
            if (!failed) {
                rc = t->xchg_gather(x.rgb, (size_t)G * ((size_t)x.hdr + (size_t)x.slots * 3 * Pa), (void*)t->stream, t->xchg_user);
                if (rc != 0) failed = "all-gather (dL_dRGB records)";
            }
            if (!failed && !overlap) {
                rc = t->xchg_reduce(x.geo, 12 * Pa, (void*)t->stream, t->xchg_user);
                if (rc != 0) failed = "all-reduce (geometry planes)";
            }
            stage.end();
            if (failed) { set_error("%s hook failed with %d", failed, rc); return GS_ERR_COLLECTIVE; }
        }
        GS_TRY(debug_check(t, 8));
        Dims d;
        GS_TRY(trainer_dims(t, &d));
        // the update rides in the rebuild (every part updates the planes it has just written): no update launch, no gradient read-back
        FusedUpdate fu;
        const bool fuse = t->opt.fuse_update != 0 && t->train.s.mean_copy != nullptr;
        if (fuse) {
            GS_TRY(prepare_adam(t, h));
            fu.u = make_update_args(Planes{ m->sh_coeffs }, *h, t->adam_t + 1);
            fu.params = m->planes; fu.am = t->adam_m.as<float>(); fu.av = t->adam_v.as<float>();
            fu.sh16 = (t->opt.sh_fp16 && t->sh16_of == (const void*)m->planes) ? t->sh16.as<uint16_t>() : nullptr;
        }
        const FusedUpdate* fup = fuse ? &fu : nullptr;
        const float* mc = t->train.s.mean_copy;
        {
            StageScope stage(t, 6, 8);
            if (overlap) {
                // the SH planes need the gathered records only: they are rebuilt while the all-reduce is still under way on the second
                // stream; the twelve other planes follow behind its event
                GS_TRY(launch_sh_rebuild(d, m->planes, mc, x, C_, per_pass, (float)t->total_samples, t->grad.as<float>(), 1, t->stream, fup));
                GS_HIP(hipStreamWaitEvent(t->stream, t->ev_reduced, 0));
                second.armed = false;   // the main stream now waits for the all-reduce itself
                GS_TRY(launch_sh_rebuild(d, m->planes, mc, x, C_, per_pass, (float)t->total_samples, t->grad.as<float>(), 2, t->stream, fup));
            } else {
                GS_TRY(launch_sh_rebuild(d, m->planes, mc, x, C_, per_pass, (float)t->total_samples, t->grad.as<float>(), 3, t->stream, fup));
            }
        }
        GS_TRY(debug_check(t, 6));
        if (fuse) {
            if (h->update_rule == GS_UPDATE_ADAM) t->adam_t++;
            if (!fu.sh16) t->sh16_of = nullptr;
            t->accumulated = false;
            update_done = true;
        }
    }
    if (!update_done) GS_TRY(apply_update(t, h, 6));
    if (densify) {   // var and the averaged location gradient are complete on every rank: densify identically, no further exchange
        GS_TRY(resolve_stats(t));
        GS_TRY(trainer_densify(t, h, &t->last));
    }
    if (stats) { GS_TRY(resolve_stats(t)); *stats = t->last; }
    return GS_OK;
}

extern "C" int gs_trainer_set_compact_exchange(gs_trainer* t, gs_collective_fn all_gather, gs_allreduce_fn all_reduce, void* user, int rank,
                                               int world, int n_cameras, const float* campos) {
    if (!t) return GS_ERR_INVALID_ARGUMENT;
    GS_HIP(hipSetDevice(t->device));
    GS_HIP(hipStreamSynchronize(t->stream));
    if (!all_gather || !all_reduce) { t->xchg_gather = nullptr; t->xchg_reduce = nullptr; t->xchg_user = nullptr; t->xchg_rank = 0; t->xchg_world = 1; t->xchg_cameras = 0; return GS_OK; }
    // `campos` is accepted for ABI compatibility and not used: every rank sends the positions of its cameras in the header of its
    // records (Exchange::hdr), taken from the view block the step rendered with, so a re-capture with moved cameras
    // (gs_trainer_set_views) needs no second call here and cannot leave a stale SH basis behind.
    (void)campos;
    if (world < 1 || world > 64 || rank < 0 || rank >= world || n_cameras < world) {
        set_error("gs_trainer_set_compact_exchange: rank %d of %d (1..64 ranks), %d cameras (at least one per rank)", rank, world, n_cameras);
        return GS_ERR_INVALID_ARGUMENT;
    }
    t->xchg_gather = all_gather; t->xchg_reduce = all_reduce; t->xchg_user = user; t->xchg_rank = rank; t->xchg_world = world; t->xchg_cameras = n_cameras;
    return GS_OK;
}

extern "C" int gs_trainer_step(gs_trainer* t, const gs_hyper* h, int densify, gs_step_stats* stats) {
    if (!t || !h) return GS_ERR_INVALID_ARGUMENT;
    // `var` (accumulateGradients, src/Trainer.cu:52) is read by the densify block of the same iteration only (:444,453)
    // and starts from zero every iteration (:304-309): a step without densify does not need per-pass gradients
    if (t->xchg_gather && t->xchg_reduce) return step_compact_exchange(t, h, densify, stats);
    // no collective between the gradients and the update: the per-splat reduction applies the update itself (north_star: "fused Adam update")
    const bool fuse_update = t->opt.fuse_update && !t->allreduce && !(t->shard_rs && t->shard_ag);
    bool update_applied = false;
    GS_TRY(accumulate_async(t, densify != 0, nullptr, fuse_update ? h : nullptr, &update_applied));
    if (update_applied) {
        if (densify) {
            GS_TRY(resolve_stats(t));
            GS_TRY(trainer_densify(t, h, &t->last));
        }
        if (stats) { GS_TRY(resolve_stats(t)); *stats = t->last; }
        return GS_OK;
    }
    if (t->shard_rs && t->shard_ag) {
        // Data-parallel sharded update: every rank ends up with the sum of its own chunk of the gradient buffer only
        // (reduce-scatter), updates that chunk of the parameters (and keeps Adam moments for that chunk only up to date),
        // and the updated chunks are exchanged (all-gather).  Same bytes on the wire as an all-reduce; the update and
        // the optimizer state are 1 / world per rank.  One chunking for all plane-major buffers (shard_total_floats).
        gs_model* m = t->model;
        const size_t n = shard_total_floats(m->sh_coeffs, m->Pa, t->shard_world), chunk = n / (size_t)t->shard_world;
        const size_t lo = chunk * (size_t)t->shard_rank, hi = lo + chunk;
        GS_TRY(shard_call(t, t->shard_rs, t->grad.as<float>(), n, "reduce-scatter", 6));
        GS_TRY(debug_check(t, 8));
        GS_TRY(apply_update(t, h, 8, lo, hi));
        GS_TRY(shard_call(t, t->shard_ag, m->planes, n, "all-gather (parameters)", 7));
        if (densify) {
            // densify reads var and the averaged location gradient of EVERY splat, and re-indexes the Adam moments:
            // complete the three buffers on every rank first (every intervalDensify-th step only)
            GS_TRY(shard_call(t, t->shard_ag, t->grad.as<float>(), n, "all-gather (gradients)", 8));
            if (t->adam_valid) {
                GS_TRY(shard_call(t, t->shard_ag, t->adam_m.as<float>(), n, "all-gather (Adam m)", 8));
                GS_TRY(shard_call(t, t->shard_ag, t->adam_v.as<float>(), n, "all-gather (Adam v)", 8));
            }
            GS_TRY(resolve_stats(t));
            GS_TRY(trainer_densify(t, h, &t->last));
        }
        if (stats) { GS_TRY(resolve_stats(t)); *stats = t->last; }
        return GS_OK;
    }
    if (t->allreduce) {
        float* buf = nullptr; size_t n = 0;
        GS_TRY(gs_trainer_grad_buffer(t, &buf, &n));
        int rc;
        prof_stage_begin(t, 8, 6);
        rc = t->allreduce(buf, n, (void*)t->stream, t->allreduce_user);
        prof_stage_end(t, 8);
        GS_TRY(debug_check(t, 8));
        if (rc != 0) { set_error("all-reduce hook failed with %d", rc); return GS_ERR_COLLECTIVE; }
    }
    return gs_trainer_apply(t, h, densify, stats);
}

extern "C" int gs_trainer_set_sharded_update(gs_trainer* t, gs_collective_fn reduce_scatter, gs_collective_fn all_gather, void* user,
                                             int rank, int world) {
    if (!t) return GS_ERR_INVALID_ARGUMENT;
    if (!reduce_scatter || !all_gather) { t->shard_rs = t->shard_ag = nullptr; t->shard_user = nullptr; t->shard_rank = 0; t->shard_world = 1; return GS_OK; }
    if (world < 1 || world > 64 || rank < 0 || rank >= world) { set_error("gs_trainer_set_sharded_update: rank %d of %d (1..64 ranks)", rank, world); return GS_ERR_INVALID_ARGUMENT; }
    GS_HIP(hipStreamSynchronize(t->stream));
    t->shard_rs = reduce_scatter; t->shard_ag = all_gather; t->shard_user = user; t->shard_rank = rank; t->shard_world = world;
    t->adam_valid = false; t->adam_t = 0;   // moments kept so far may be stale outside a chunk: start clean
    return GS_OK;
}

extern "C" int gs_trainer_adam_state(gs_trainer* t, float** m, float** v, size_t* n_floats, int* steps) {
    if (!t || !t->model) return GS_ERR_INVALID_ARGUMENT;
    GS_HIP(hipStreamSynchronize(t->stream));
    const bool ok = t->adam_valid;
    if (m) *m = ok ? t->adam_m.as<float>() : nullptr;
    if (v) *v = ok ? t->adam_v.as<float>() : nullptr;
    if (n_floats) *n_floats = ok ? (size_t)(11 + 3 * t->model->sh_coeffs) * t->model->Pa : 0;
    if (steps) *steps = ok ? t->adam_t : 0;
    return GS_OK;
}

// Checkpoint / resume of an Adam run: the caller hands back what gs_trainer_adam_state gave out (device or host pointers,
// n_floats = (11 + 3M) * plane stride of the CURRENT model) and the step counter.  m1 == m2 == NULL with steps == 0 resets.
extern "C" int gs_trainer_set_adam_state(gs_trainer* t, const float* m1, const float* m2, size_t n_floats, int steps, int on_device) {
    if (!t || !t->model) return GS_ERR_INVALID_ARGUMENT;
    GS_HIP(hipSetDevice(t->device));
    GS_HIP(hipStreamSynchronize(t->stream));
    if (!m1 && !m2 && steps == 0) { t->adam_valid = false; t->adam_t = 0; return GS_OK; }
    gs_model* m = t->model;
    const size_t want = (size_t)(11 + 3 * m->sh_coeffs) * m->Pa;
    if (!m1 || !m2 || steps < 0 || n_floats != want) {
        set_error("gs_trainer_set_adam_state: expected two moment buffers of %zu floats ((11 + 3M) x plane stride of the current model) and steps >= 0, got %zu floats, steps %d",
                  want, n_floats, steps);
        return GS_ERR_INVALID_ARGUMENT;
    }
    if (t->shard_rs && t->shard_world > 1) {
        set_error("gs_trainer_set_adam_state: under the sharded update every rank holds the moments of its own chunk only; restore before gs_trainer_set_sharded_update is not kept either (it resets the moments)");
        return GS_ERR_INVALID_ARGUMENT;
    }
    const size_t bytes = plane_buffer_floats(m->sh_coeffs, m->Pa) * 4;
    GS_TRY(t->adam_m.ensure(bytes)); GS_TRY(t->adam_v.ensure(bytes));
    GS_HIP(hipMemsetAsync(t->adam_m.p, 0, bytes, t->stream));
    GS_HIP(hipMemsetAsync(t->adam_v.p, 0, bytes, t->stream));
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    GS_HIP(hipMemcpyAsync(t->adam_m.p, m1, want * 4, kind, t->stream));
    GS_HIP(hipMemcpyAsync(t->adam_v.p, m2, want * 4, kind, t->stream));
    GS_HIP(hipStreamSynchronize(t->stream));  // host sources may be released on return
    t->adam_valid = true; t->adam_t = steps;
    return GS_OK;
}

extern "C" int gs_trainer_set_option(gs_trainer* t, const char* name, int value) {
    if (!t || !name) return GS_ERR_INVALID_ARGUMENT;
    const int share_before = t->opt.share;
    if (!set_option(t->opt, name, value)) { set_error("gs_trainer_set_option: unknown option '%s'", name); return GS_ERR_INVALID_ARGUMENT; }
    if (t->opt.share != share_before && t->V > 0) {  // regroup the passes already set
        t->VG = build_view_block(t->h_views.data(), t->V, t->opt.share != 0, t->h_view_block, &t->bwd_pairs, &t->bwd_singles);
        const int* vg = reinterpret_cast<const int*>(t->h_view_block.data() + (size_t)t->V * 2 * sizeof(gs_view));
        t->h_view_group.assign(vg, vg + t->V);
        t->views_on_device = nullptr;
        // the group count changed: the host's hint words (h_flags[g * 4 + 1 / 3]) belong to other groups, and gradients
        // accumulated under the old grouping must not be applied (as in gs_trainer_set_views)
        t->steps_on_these_lists = 0; t->accumulated = false; t->stats_stale = false;
    }
    if (strcmp(name, "sh_fp16") == 0) t->sh16_of = nullptr;
    return GS_OK;
}

extern "C" int gs_trainer_set_allreduce(gs_trainer* t, gs_allreduce_fn fn, void* user) {
    if (!t) return GS_ERR_INVALID_ARGUMENT;
    t->allreduce = fn; t->allreduce_user = user;
    return GS_OK;
}
extern "C" int gs_trainer_get_stream(gs_trainer* t, void** s) {
    if (!t || !s) return GS_ERR_INVALID_ARGUMENT;
    *s = (void*)t->stream;
    return GS_OK;
}
extern "C" int gs_trainer_synchronize(gs_trainer* t) {
    if (!t) return GS_ERR_INVALID_ARGUMENT;
    GS_HIP(hipStreamSynchronize(t->stream));
    return GS_OK;
}

extern "C" int gs_trainer_set_profiling(gs_trainer* t, int enable) {
    if (!t) return GS_ERR_INVALID_ARGUMENT;
    GS_HIP(hipStreamSynchronize(t->stream));
    prof_resolve(t);
    t->profiling = enable != 0;
    t->profiling_mask = enable == 1 ? ~0u : (unsigned)enable >> 1;
    t->prev_mark = nullptr; t->stage_start = nullptr;
    for (int i = 0; i < GS_STAGE_COUNT; i++) { t->stage_ms[i] = 0; t->stage_launches[i] = 0; }
    return GS_OK;
}
extern "C" int gs_trainer_stage_times(gs_trainer* t, double* ms_sum, long long* launches) {
    if (!t) return GS_ERR_INVALID_ARGUMENT;
    GS_HIP(hipStreamSynchronize(t->stream));
    prof_resolve(t);
    for (int i = 0; i < GS_STAGE_COUNT; i++) { if (ms_sum) ms_sum[i] = t->stage_ms[i]; if (launches) launches[i] = t->stage_launches[i]; }
    return GS_OK;
}
extern "C" const char* gs_stage_name(int i) { return (i >= 0 && i < GS_STAGE_COUNT) ? kStageNames[i] : ""; }

extern "C" int gs_trainer_read_image(gs_trainer* t, int view, float* host_chw) {
    if (!t || !host_chw || view < 0 || view >= t->V || !t->train.s.out_color) return GS_ERR_INVALID_ARGUMENT;
    GS_HIP(hipStreamSynchronize(t->stream));
    const size_t N = (size_t)t->W * t->H;
    GS_HIP(hipMemcpy(host_chw, t->train.s.out_color + (size_t)view * 3 * N, 3 * N * 4, hipMemcpyDeviceToHost));
    return GS_OK;
}

extern "C" int gs_trainer_render(gs_trainer* t, uint32_t* fb, int fb_on_device, int w, int h, float splat_scale, const gs_view* view) {
    if (!t || !fb || !view || w <= 0 || h <= 0) return GS_ERR_INVALID_ARGUMENT;
    if (!t->model) return GS_ERR_NO_MODEL;
    GS_HIP(hipSetDevice(t->device));
    gs_model* m = t->model;
    int D = 0;
    GS_TRY(effective_degree(m->sh_degree, m->sh_coeffs, &D));
    uint32_t rcap = std::max<uint32_t>(t->preview.Rcap, (uint32_t)std::max<long long>(1 << 20, 16LL * m->count));
    DevBuf fbdev;
    for (int attempt = 0;; attempt++) {
        GS_TRY(t->preview.ensure(m->count, 1, w, h, rcap));
        Dims d = make_dims(m->count, m->Pa, D, m->sh_coeffs, w, h, 1, t->preview.Rcap, splat_scale, -1, t->opt.cull);
        Scratch s = t->preview.s;
        std::vector<char> vb;
        build_view_block(view, 1, false, vb);
        GS_HIP(hipMemcpyAsync((void*)s.views, vb.data(), vb.size(), hipMemcpyHostToDevice, t->stream));
        GS_HIP(hipStreamSynchronize(t->stream));  // vb is a stack-lifetime staging buffer
        GS_TRY(stage_project(d, m->planes, s, t->stream));  // an empty model gives empty lists and the background image
        GS_TRY(stage_bin_render(d, s, t->stream));
        uint32_t flags[4];
        GS_HIP(hipMemcpyAsync(flags, s.flags, 16, hipMemcpyDeviceToHost, t->stream));
        GS_HIP(hipStreamSynchronize(t->stream));
        if (!(flags[0] & 1u)) break;
        if (attempt > 8) { set_error("preview arena failed to converge"); return GS_ERR_INTERNAL; }
        rcap = flags[2] + flags[2] / 4 + 1024;
    }
    uint32_t* target = fb;
    const size_t N = (size_t)w * h;
    if (!fb_on_device) { GS_TRY(fbdev.ensure(N * 4)); target = fbdev.as<uint32_t>(); }
    int rc = launch_image_float_to_int(t->preview.s.out_color, target, w, h, t->stream);
    if (rc == GS_OK && hipStreamSynchronize(t->stream) != hipSuccess) rc = GS_ERR_HIP;  // cudaDeviceSynchronize, src/Trainer.cu:215
    if (rc == GS_OK && !fb_on_device && hipMemcpy(fb, target, N * 4, hipMemcpyDeviceToHost) != hipSuccess) rc = GS_ERR_HIP;
    fbdev.release();
    if (rc == GS_ERR_HIP) set_error("render failed: %s", hipGetErrorString(hipGetLastError()));
    return rc;
}

// =============================================================================================
// image kernels as entry points
// =============================================================================================
extern "C" int gs_image_float_to_int(const float* src, uint32_t* fb, int w, int h) {
    if (!src || !fb || w <= 0 || h <= 0) { set_error("gs_image_float_to_int: NULL image or empty size (%d x %d)", w, h); return GS_ERR_INVALID_ARGUMENT; }
    GS_TRY(launch_image_float_to_int(src, fb, w, h, 0));
    GS_HIP(hipDeviceSynchronize());
    return GS_OK;
}
extern "C" int gs_image_int_to_loss(const uint32_t* truth, const float* rast, float* loss, int w, int h) {
    if (!truth || !rast || !loss || w <= 0 || h <= 0) { set_error("gs_image_int_to_loss: NULL image or empty size (%d x %d)", w, h); return GS_ERR_INVALID_ARGUMENT; }
    GS_TRY(launch_image_int_to_loss(truth, rast, loss, w, h, 0));
    GS_HIP(hipDeviceSynchronize());
    return GS_OK;
}

// =============================================================================================
// rasterizer seam
// =============================================================================================
namespace {
struct Field { const char* name; size_t off, bytes; };
inline size_t al(size_t x) { return round_up_sz(x, 256); }

struct GeomLayout { size_t record, tiles, offsets, scan_tmp, wg_hist, view, flags, planes, total; int Pa; };
GeomLayout geom_layout(int P, int M, int W = 16, int H = 16) {
    GeomLayout L; L.Pa = std::max(64, round_up(P, 64));
    size_t o = 0;
    L.record = o; o = al(o + (size_t)L.Pa * sizeof(GeomRec));
    L.tiles = o; o = al(o + (size_t)L.Pa * 4);
    L.offsets = o; o = al(o + (size_t)L.Pa * 4);
    {
        const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
        const int T = gx * gy, NST = ((gx + STILE - 1) / STILE) * ((gy + STILE - 1) / STILE);
        L.scan_tmp = o; o = al(o + (splat_blocks(L.Pa) + scan_partials_count(std::max(T, NST), 1) + 64) * 4);
        L.wg_hist = o; o = al(o + (size_t)splat_blocks(L.Pa) * NST * 4);
    }
    L.view = o; o = al(o + view_block_bytes(1));
    L.flags = o; o = al(o + 32);
    L.planes = o; o = al(o + (size_t)(11 + 3 * M) * L.Pa * 4);
    L.total = o;
    return L;
}
struct ImageLayout { size_t coarse_count, coarse_end, tile_count, tile_end, tile_order, ranges, final_T, n_contrib, scan_tmp, total; int T, NST; };
ImageLayout image_layout(int W, int H) {
    ImageLayout L; const size_t N = (size_t)W * H;
    const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
    L.T = gx * gy;
    L.NST = ((gx + STILE - 1) / STILE) * ((gy + STILE - 1) / STILE);
    size_t o = 0;
    L.coarse_count = o; o = al(o + (size_t)L.NST * 4);
    L.coarse_end = o; o = al(o + (size_t)L.NST * 4);
    L.tile_count = o; o = al(o + (size_t)L.T * 4);
    L.tile_end = o; o = al(o + (size_t)L.T * 4);
    L.tile_order = o; o = al(o + (size_t)L.T * 4);
    L.ranges = o; o = al(o + (size_t)L.T * 8);
    L.final_T = o; o = al(o + N * 4);
    L.n_contrib = o; o = al(o + N * 4);
    L.scan_tmp = o; o = al(o + (scan_partials_count(L.T, 1) + scan_partials_count(L.NST, 1) + 64) * 4);
    L.total = o;
    return L;
}
struct BinLayout { size_t clist, cdepth, ids, plist, slist, G, hmask, marks, total; uint32_t Rcap; };
BinLayout bin_layout(int R, int T) {
    BinLayout L; L.Rcap = (uint32_t)std::max(R, 1);
    size_t o = 0;
    L.clist = o; o = al(o + (size_t)L.Rcap * 16);
    L.cdepth = o; o = al(o + (size_t)L.Rcap * 4);
    L.ids = o; o = al(o + (size_t)L.Rcap * 4);
    L.plist = o; o = al(o + (size_t)L.Rcap * 4);
    L.slist = o; o = al(o + (size_t)L.Rcap * 4);
    L.G = o; o = al(o + (size_t)L.Rcap * G_STRIDE * 4);
    L.hmask = o; o = al(o + hit_mask_words(L.Rcap, T) * 4 * sizeof(unsigned long long));  // (appended: the offsets above are part of the seam)
    L.marks = o; o = al(o + 64 + (size_t)L.Rcap + 64);  // 64 zero bytes (Scratch::zero_row), then which slots hold a gradient row (Scratch::row_epoch)
    L.total = o;
    return L;
}
Scratch seam_scratch(char* geom, const GeomLayout& g, char* img, const ImageLayout& im, char* bin, const BinLayout* b) {
    Scratch s{};
    set_view_block_pointers(s, geom + g.view, 1);
    s.geom = reinterpret_cast<GeomRec*>(geom + g.record);
    s.tiles_touched = reinterpret_cast<uint32_t*>(geom + g.tiles);
    s.point_offsets = reinterpret_cast<uint32_t*>(geom + g.offsets);
    s.block_sums = reinterpret_cast<uint32_t*>(geom + g.scan_tmp);
    s.flags = reinterpret_cast<uint32_t*>(geom + g.flags);
    s.sort_marks = s.flags + 5;  // the block holds 8 words: flags[4], a loss total, these two
    s.coarse_count = reinterpret_cast<uint32_t*>(img + im.coarse_count);
    s.wg_hist = reinterpret_cast<uint32_t*>(geom + g.wg_hist);
    s.coarse_end = reinterpret_cast<uint32_t*>(img + im.coarse_end);
    s.tile_count = reinterpret_cast<uint32_t*>(img + im.tile_count);
    s.tile_end = reinterpret_cast<uint32_t*>(img + im.tile_end);
    s.tile_order = reinterpret_cast<uint32_t*>(img + im.tile_order);
    s.final_T = reinterpret_cast<float*>(img + im.final_T);
    s.n_contrib = reinterpret_cast<uint32_t*>(img + im.n_contrib);
    if (bin && b) {
        s.coarse_list = reinterpret_cast<uint4*>(bin + b->clist);
        s.coarse_depth = reinterpret_cast<uint32_t*>(bin + b->cdepth);
        s.id_of_slot = reinterpret_cast<uint32_t*>(bin + b->ids);
        s.point_list = reinterpret_cast<uint32_t*>(bin + b->plist);
        s.slot_list = reinterpret_cast<uint32_t*>(bin + b->slist);
        s.G = reinterpret_cast<float*>(bin + b->G);
        s.hit_masks = reinterpret_cast<unsigned long long*>(bin + b->hmask);
        s.zero_row = reinterpret_cast<const float*>(bin + b->marks);
        s.row_epoch = reinterpret_cast<uint8_t*>(bin + b->marks) + 64;
    }
    return s;
}
}  // namespace

extern "C" int gs_raster_chunk_field(const char* chunk, const char* field, int P, int width, int height, int R,
                                     size_t* offset, size_t* bytes) {
    if (!chunk || !field || !offset || !bytes) return GS_ERR_INVALID_ARGUMENT;
    const std::string c(chunk), f(field);
    if (c == "geometry") {
        const GeomLayout L = geom_layout(P, 1, width, height);
        if (f == "record") { *offset = L.record; *bytes = (size_t)P * sizeof(GeomRec); return GS_OK; }
        if (f == "tiles_touched") { *offset = L.tiles; *bytes = (size_t)P * 4; return GS_OK; }
        if (f == "point_offsets") { *offset = L.offsets; *bytes = (size_t)P * 4; return GS_OK; }
    } else if (c == "image") {
        const ImageLayout L = image_layout(width, height);
        const size_t N = (size_t)width * height;
        if (f == "ranges") { *offset = L.ranges; *bytes = (size_t)L.T * 8; return GS_OK; }
        if (f == "tile_order") { *offset = L.tile_order; *bytes = (size_t)L.T * 4; return GS_OK; }
        if (f == "final_T") { *offset = L.final_T; *bytes = N * 4; return GS_OK; }
        if (f == "n_contrib") { *offset = L.n_contrib; *bytes = N * 4; return GS_OK; }
    } else if (c == "binning") {
        const BinLayout L = bin_layout(R, image_layout(width, height).T);
        if (f == "point_list") { *offset = L.plist; *bytes = (size_t)R * 4; return GS_OK; }
        if (f == "point_list_slots") { *offset = L.slist; *bytes = (size_t)R * 4; return GS_OK; }
    }
    set_error("gs_raster_chunk_field: unknown %s/%s", chunk, field);
    return GS_ERR_INVALID_ARGUMENT;
}

extern "C" int gs_rasterize_forward(gs_alloc_fn geometry_alloc, void* geometry_user, gs_alloc_fn binning_alloc,
                                    void* binning_user, gs_alloc_fn image_alloc, void* image_user, int P, int D_in, int M,
                                    const float* background, int width, int height, const float* means3D, const float* shs,
                                    const float* colors_precomp, const float* opacities, const float* scales,
                                    float scale_modifier, const float* rotations, const float* cov3D_precomp,
                                    const float* viewmatrix, const float* projmatrix, const float* cam_pos, float tan_fovx,
                                    float tan_fovy, int prefiltered, float* out_color, int* radii, int debug,
                                    int* num_rendered) {
    (void)debug;
    if (!geometry_alloc || !binning_alloc || !image_alloc || P < 0 || width <= 0 || height <= 0 || !out_color || !background ||
        !viewmatrix || !projmatrix || !cam_pos || (P > 0 && (!means3D || !shs || !opacities || !scales || !rotations))) {
        set_error("gs_rasterize_forward: bad arguments");
        return GS_ERR_INVALID_ARGUMENT;
    }
    if (colors_precomp || cov3D_precomp || radii || prefiltered) {
        set_error("gs_rasterize_forward: colors_precomp / cov3D_precomp / radii / prefiltered are not used by the reference and not supported");
        return GS_ERR_INVALID_ARGUMENT;
    }
    GS_TRY(require_device());
    int D = 0;
    GS_TRY(effective_degree(D_in, M, &D));
    const GeomLayout gl = geom_layout(P, M, width, height);
    const ImageLayout il = image_layout(width, height);
    char* geom = geometry_alloc(gl.total, geometry_user);
    char* img = image_alloc(il.total, image_user);
    if (!geom || !img) { set_error("allocator callback returned null"); return GS_ERR_OUT_OF_MEMORY; }
    hipStream_t st = 0;
    Scratch s = seam_scratch(geom, gl, img, il, nullptr, nullptr);
    s.out_color = out_color;
    // view block (pass 0 = camera 0) assembled on the device from the caller's device pointers
    {
        gs_view zero{};
        std::vector<char> host;
        build_view_block(&zero, 1, false, host);
        GS_HIP(hipMemcpyAsync(geom + gl.view, host.data(), host.size(), hipMemcpyHostToDevice, st));
        GS_HIP(hipStreamSynchronize(st));
        for (int copy = 0; copy < 2; copy++) {  // the pass's view and its group's camera are the same struct here
            char* vb = geom + gl.view + copy * sizeof(gs_view);
            GS_HIP(hipMemcpyAsync(vb + offsetof(gs_view, view), viewmatrix, 64, hipMemcpyDeviceToDevice, st));
            GS_HIP(hipMemcpyAsync(vb + offsetof(gs_view, projview), projmatrix, 64, hipMemcpyDeviceToDevice, st));
            GS_HIP(hipMemcpyAsync(vb + offsetof(gs_view, campos), cam_pos, 12, hipMemcpyDeviceToDevice, st));
            const float tf[2] = { tan_fovx, tan_fovy };
            GS_HIP(hipMemcpyAsync(vb + offsetof(gs_view, tan_fovx), tf, 8, hipMemcpyHostToDevice, st));
            GS_HIP(hipMemcpyAsync(vb + offsetof(gs_view, bg), background, 12, hipMemcpyDeviceToDevice, st));
        }
    }
    float* planes = reinterpret_cast<float*>(geom + gl.planes);
    GS_TRY(launch_aos_to_soa(P, gl.Pa, M, means3D, shs, scales, opacities, rotations, planes, st));
    Dims d = make_dims(P, gl.Pa, D, M, width, height, 1, 0xFFFFFFFFu, scale_modifier);  // no arena yet: nothing can overflow
    int R = 0;
    GS_TRY(stage_project(d, planes, s, st));
    if (P > 0) {
        // the reference reads num_rendered back here too (a device->host sync)
        uint32_t r32 = 0;
        GS_HIP(hipMemcpyAsync(&r32, s.flags + 2, 4, hipMemcpyDeviceToHost, st));
        GS_HIP(hipStreamSynchronize(st));
        if (r32 >= (1u << 31) - 64u) { set_error("gs_rasterize_forward: %u (splat, tile) entries exceed what the int num_rendered of the interface holds", r32); return GS_ERR_OUT_OF_MEMORY; }
        R = (int)r32;
    }
    const BinLayout bl = bin_layout(R, il.T);
    char* bin = binning_alloc(bl.total, binning_user);
    if (!bin) { set_error("binning allocator returned null"); return GS_ERR_OUT_OF_MEMORY; }
    s = seam_scratch(geom, gl, img, il, bin, &bl);
    s.out_color = out_color;
    d.Rcap = bl.Rcap;
    GS_HIP(hipMemsetAsync(bin + bl.marks, 0, 64 + (size_t)bl.Rcap + 64, st));  // the caller's chunk is uninitialised; the backward marks its rows with epoch 1
    GS_TRY(stage_bin_render(d, s, st));  // P == 0: empty lists, background image
    GS_TRY(launch_ranges(d, s, reinterpret_cast<uint32_t*>(img + il.ranges), st));
    GS_HIP(hipStreamSynchronize(st));
    if (num_rendered) *num_rendered = R;
    return GS_OK;
}

extern "C" int gs_rasterize_backward(int P, int D_in, int M, int R, const float* background, int width, int height,
                                     const float* means3D, const float* shs, const float* colors_precomp,
                                     const float* scales, float scale_modifier, const float* rotations,
                                     const float* cov3D_precomp, const float* viewmatrix, const float* projmatrix,
                                     const float* campos, float tan_fovx, float tan_fovy, const int* radii,
                                     char* geom_buffer, char* binning_buffer, char* image_buffer, const float* dL_dpix,
                                     float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolor,
                                     float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot,
                                     int debug) {
    (void)debug; (void)background; (void)viewmatrix; (void)projmatrix; (void)campos; (void)tan_fovx; (void)tan_fovy;
    (void)means3D; (void)shs; (void)scales; (void)rotations;  // the geometry chunk already holds all of them
    if (P < 0 || R < 0 || !geom_buffer || !binning_buffer || !image_buffer || !dL_dpix || !dL_dmean2D || !dL_dconic ||
        !dL_dopacity || !dL_dcolor || !dL_dmean3D || !dL_dcov3D || !dL_dsh || !dL_dscale || !dL_drot) {
        set_error("gs_rasterize_backward: bad arguments");
        return GS_ERR_INVALID_ARGUMENT;
    }
    if (colors_precomp || cov3D_precomp || radii) {
        set_error("gs_rasterize_backward: colors_precomp / cov3D_precomp / radii must be NULL as at the reference call site");
        return GS_ERR_INVALID_ARGUMENT;
    }
    int D = 0;
    GS_TRY(effective_degree(D_in, M, &D));
    if (P == 0) return GS_OK;
    const GeomLayout gl = geom_layout(P, M, width, height);
    const ImageLayout il = image_layout(width, height);
    const BinLayout bl = bin_layout(R, il.T);
    Scratch s = seam_scratch(geom_buffer, gl, image_buffer, il, binning_buffer, &bl);
    s.dL_dpix = dL_dpix;
    Dims d = make_dims(P, gl.Pa, D, M, width, height, 1, bl.Rcap, scale_modifier);
    hipStream_t st = 0;
    GS_TRY(launch_render_backward(d, s, reinterpret_cast<const int*>(geom_buffer + gl.view + view_block_items_offset(1)), 0, 1, false, st));
    SeamGrads g{ dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot };
    GS_TRY(launch_splat_backward_seam(d, reinterpret_cast<const float*>(geom_buffer + gl.planes), s, g, st));
    GS_HIP(hipStreamSynchronize(st));
    return GS_OK;
}

// =============================================================================================
// RCCL communicator (librccl resolved lazily so that single-GPU users never load it)
// =============================================================================================
namespace {
struct RcclId { char internal[128]; };
typedef int (*fn_get_id)(RcclId*);
typedef int (*fn_init_rank)(void**, int, RcclId, int);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_reduce_scatter)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_all_gather)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*fn_destroy)(void*);
typedef int (*fn_abort)(void*);
typedef const char* (*fn_errstr)(int);
struct Rccl {
    void* lib = nullptr;
    fn_get_id get_id = nullptr; fn_init_rank init_rank = nullptr; fn_allreduce allreduce = nullptr; fn_destroy destroy = nullptr;
    fn_reduce_scatter reduce_scatter = nullptr; fn_all_gather all_gather = nullptr;
    fn_errstr errstr = nullptr;
    fn_abort abort = nullptr;
} g_rccl;
int rccl_load() {
    if (g_rccl.lib) return GS_OK;
    void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) { set_error("cannot load librccl: %s", dlerror()); return GS_ERR_INTERNAL; }
    g_rccl.get_id = (fn_get_id)dlsym(lib, "ncclGetUniqueId");
    g_rccl.init_rank = (fn_init_rank)dlsym(lib, "ncclCommInitRank");
    g_rccl.allreduce = (fn_allreduce)dlsym(lib, "ncclAllReduce");
    g_rccl.destroy = (fn_destroy)dlsym(lib, "ncclCommDestroy");
    g_rccl.reduce_scatter = (fn_reduce_scatter)dlsym(lib, "ncclReduceScatter");
    g_rccl.all_gather = (fn_all_gather)dlsym(lib, "ncclAllGather");
    g_rccl.errstr = (fn_errstr)dlsym(lib, "ncclGetErrorString");
    g_rccl.abort = (fn_abort)dlsym(lib, "ncclCommAbort");
    if (!g_rccl.get_id || !g_rccl.init_rank || !g_rccl.allreduce || !g_rccl.destroy) { set_error("librccl lacks expected symbols"); return GS_ERR_INTERNAL; }
    g_rccl.lib = lib;
    return GS_OK;
}
}  // namespace

struct gs_comm { void* comm = nullptr; int rank = 0, n_ranks = 1; };

extern "C" int gs_comm_unique_id(char id[GS_COMM_ID_BYTES]) {
    if (!id) return GS_ERR_INVALID_ARGUMENT;
    GS_TRY(rccl_load());
    RcclId u;
    const int rc = g_rccl.get_id(&u);
    if (rc != 0) { set_error("ncclGetUniqueId failed: %s", g_rccl.errstr ? g_rccl.errstr(rc) : "?"); return GS_ERR_INTERNAL; }
    memcpy(id, u.internal, GS_COMM_ID_BYTES);
    return GS_OK;
}
extern "C" int gs_comm_create(const char id[GS_COMM_ID_BYTES], int rank, int n_ranks, gs_comm** out) {
    if (!id || !out || rank < 0 || n_ranks < 1 || rank >= n_ranks) return GS_ERR_INVALID_ARGUMENT;
    GS_TRY(rccl_load());
    RcclId u;
    memcpy(u.internal, id, GS_COMM_ID_BYTES);
    gs_comm* c = new gs_comm();
    c->rank = rank; c->n_ranks = n_ranks;
    const int rc = g_rccl.init_rank(&c->comm, n_ranks, u, rank);
    if (rc != 0) { delete c; set_error("ncclCommInitRank failed: %s", g_rccl.errstr ? g_rccl.errstr(rc) : "?"); return GS_ERR_INTERNAL; }
    *out = c;
    return GS_OK;
}
extern "C" int gs_comm_destroy(gs_comm* c) {
    if (!c) return GS_OK;
    if (c->comm && g_rccl.destroy) g_rccl.destroy(c->comm);
    delete c;
    return GS_OK;
}
// A collective that fails on this rank aborts the communicator (ncclCommAbort): the peers' pending collectives then end with
// an error instead of waiting for this rank for ever.
static int rccl_result(gs_comm* c, int rc, const char* what) {
    if (rc == 0) return 0;
    set_error("%s failed: %s; communicator aborted", what, g_rccl.errstr ? g_rccl.errstr(rc) : "?");
    if (c->comm && g_rccl.abort) { g_rccl.abort(c->comm); c->comm = nullptr; }
    return rc;
}
static int rccl_hook(float* buf, size_t n, void* stream, void* user) {
    gs_comm* c = static_cast<gs_comm*>(user);
    if (!c->comm) return -1;
    // ncclFloat32 = 7, ncclSum = 0
    return rccl_result(c, g_rccl.allreduce(buf, buf, n, 7, 0, c->comm, (hipStream_t)stream), "ncclAllReduce");
}
extern "C" int gs_trainer_attach_comm(gs_trainer* t, gs_comm* c) {
    if (!t) return GS_ERR_INVALID_ARGUMENT;
    if (!c) return gs_trainer_set_allreduce(t, nullptr, nullptr);
    return gs_trainer_set_allreduce(t, rccl_hook, c);
}
// in-place forms: the rank's chunk of the buffer is both the reduce-scatter's output and the all-gather's input
static int rccl_rs_hook(float* buf, size_t n, void* stream, void* user) {
    gs_comm* c = static_cast<gs_comm*>(user);
    if (!c->comm) return -1;
    const size_t chunk = n / (size_t)c->n_ranks;
    return rccl_result(c, g_rccl.reduce_scatter(buf, buf + chunk * (size_t)c->rank, chunk, 7, 0, c->comm, (hipStream_t)stream), "ncclReduceScatter");
}
static int rccl_ag_hook(float* buf, size_t n, void* stream, void* user) {
    gs_comm* c = static_cast<gs_comm*>(user);
    if (!c->comm) return -1;
    const size_t chunk = n / (size_t)c->n_ranks;
    return rccl_result(c, g_rccl.all_gather(buf + chunk * (size_t)c->rank, buf, chunk, 7, c->comm, (hipStream_t)stream), "ncclAllGather");
}
// Compact exchange over the library's own communicators.  The all-reduce of the geometry planes is issued on the trainer's second
// stream beside the all-gather: collectives of ONE communicator execute in issue order whatever their streams, so the overlap
// needs a communicator of its own (`reduce`; the same handle as `gather` is accepted and serialises the two).
static int rccl_x_gather(float* buf, size_t n, void* stream, void* user) { return rccl_ag_hook(buf, n, stream, static_cast<void**>(user)[0]); }
static int rccl_x_reduce(float* buf, size_t n, void* stream, void* user) { return rccl_hook(buf, n, stream, static_cast<void**>(user)[1]); }
extern "C" int gs_trainer_attach_comm_compact(gs_trainer* t, gs_comm* gather, gs_comm* reduce, int n_cameras, const float* campos) {
    if (!t) return GS_ERR_INVALID_ARGUMENT;
    if (!gather) return gs_trainer_set_compact_exchange(t, nullptr, nullptr, nullptr, 0, 1, 0, nullptr);
    if (!reduce) reduce = gather;
    if (!g_rccl.all_gather) { set_error("librccl lacks ncclAllGather"); return GS_ERR_INTERNAL; }
    if (gather->rank != reduce->rank || gather->n_ranks != reduce->n_ranks) { set_error("gs_trainer_attach_comm_compact: the two communicators differ in rank / size"); return GS_ERR_INVALID_ARGUMENT; }
    t->xchg_comms[0] = gather; t->xchg_comms[1] = reduce;
    return gs_trainer_set_compact_exchange(t, rccl_x_gather, rccl_x_reduce, t->xchg_comms, gather->rank, gather->n_ranks, n_cameras, campos);
}
extern "C" int gs_trainer_attach_comm_sharded(gs_trainer* t, gs_comm* c) {
    if (!t) return GS_ERR_INVALID_ARGUMENT;
    if (!c) return gs_trainer_set_sharded_update(t, nullptr, nullptr, nullptr, 0, 1);
    if (!g_rccl.reduce_scatter || !g_rccl.all_gather) { set_error("librccl lacks ncclReduceScatter / ncclAllGather"); return GS_ERR_INTERNAL; }
    return gs_trainer_set_sharded_update(t, rccl_rs_hook, rccl_ag_hook, c, c->rank, c->n_ranks);
}
