// gs_internal.h — shared declarations of libgsplat_mi355.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/gsplat.h"

namespace gs {

// ---------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------
void set_error(const char* fmt, ...);
#define GS_HIP(expr)                                                                           \
    do {                                                                                       \
        hipError_t e__ = (expr);                                                               \
        if (e__ != hipSuccess) {                                                               \
            gs::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return GS_ERR_HIP;                                                                 \
        }                                                                                      \
    } while (0)
#define GS_TRY(expr)            \
    do {                        \
        int s__ = (expr);       \
        if (s__ != GS_OK) return s__; \
    } while (0)

// ---------------------------------------------------------------------------------------------
// layout constants
// ---------------------------------------------------------------------------------------------
constexpr int TILE = 16;           // 16x16-pixel tiles (upstream BLOCK_X/BLOCK_Y)
constexpr int WG = 256;            // workgroup = 4 wave64
constexpr int G_STRIDE = 9;        // floats per (splat, tile, pass) gradient row: 36 bytes, three 12-byte groups (no padding:
                                   // the rows are the step's largest HBM stream, written once and read once)
struct alignas(4) Row3 { float a, b, c; };  // one 12-byte group of a gradient row
#ifndef GS_SORT_TILE_NT
#define GS_SORT_TILE_NT 128              // threads and LDS capacity of the short-list sort (tuning hooks, tools/build_variant.sh)
#define GS_SORT_TILE_CAP 512
#endif
constexpr int SORT_TINY_CAP = GS_SORT_TILE_CAP;  // lists shorter than this: one 128-thread workgroup per tile, 7 KB LDS (22 tiles in flight per CU)
constexpr int SORT_SMALL_CAP = 2048;  // lists from there up to this: 256-thread workgroups walking the head of the tile order, 24 KB LDS
// flags[g * 4 + 3] carries two positions of the tile order to the host as 16-bit counts of this many tiles (hints, rounded)
inline __host__ __device__ uint32_t order_hint_unit(int T) { return (uint32_t)T / 65535u + 1u; }
constexpr int SORT_LDS_CAP = 8192;    // longest list sorted in LDS (long-list kernel, 96 KB); beyond: global scratch
constexpr int STILE = 4;           // a super-tile is STILE x STILE tiles (64x64 px): the coarse binning unit
constexpr int MAX_SUPER_TILES = 16384;  // the binning keeps one LDS counter per super-tile (64 KB at most): images up to 8192 x 8192 —
                                        // the largest render the reference's UI offers (src/ui/tools/UiPanelToolsView.cpp:120,125) —
                                        // or any other shape of as many 64 x 64-px blocks (16384 x 4096, ...)

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
inline size_t round_up_sz(size_t x, size_t m) { return (x + m - 1) / m * m; }

// Parameter planes of the SoA model, reference order (loc, sh, scale, opacity, rot); the gradient
// buffer appends one `var` plane.  plane p of splat i lives at base[p * stride + i].
struct Planes {
    int M;
    __host__ __device__ int loc(int c) const { return c; }
    __host__ __device__ int sh(int k, int c) const { return 3 + k * 3 + c; }
    __host__ __device__ int scale(int c) const { return 3 + 3 * M + c; }
    __host__ __device__ int opac() const { return 6 + 3 * M; }
    __host__ __device__ int rot(int c) const { return 7 + 3 * M + c; }
    __host__ __device__ int count() const { return 11 + 3 * M; }
    __host__ __device__ int var() const { return 11 + 3 * M; }  // only in gradient buffers
};

// 64-byte per-(view,splat) record written by preprocess, gathered by the render kernels.
struct alignas(16) GeomRec {
    float x, y, conA, conB;        // means2D, conic.x, conic.y
    float conC, opacity, r, g;     // conic.z, opacity, rgb.r, rgb.g
    float b, hx, hy, depth;        // rgb.b, hx = cull threshold tau = ln(255*opacity)+margin (hy unused), view depth
    int radius;                    // 0 = culled
    uint32_t flags;                // bits 0..2: clamped r,g,b
    uint32_t rect_min, rect_max;   // x | y << 16
};
static_assert(sizeof(GeomRec) == 64, "GeomRec must be one 64-byte line");

struct Dims {
    int P;        // splats
    int Pa;       // plane stride (multiple of 64)
    int D;        // active SH degree 0..3
    int M;        // SH coefficients per channel (plane count)
    int W, H, N;  // image
    int gx, gy, T;
    int sgx, sgy, NST;  // super-tile grid
    int V;        // views (passes) in this launch
    int VG;       // geometry groups: passes with bit-identical camera parameters share projection, lists and blend
    uint32_t Rcap;  // entries per view the binning arena holds
    float mod;    // scale modifier
    int cull;     // 1: sub-tile alpha>=1/255 box culling on (default); 0: evaluate every staged pair
    int long_sort;  // 1: k_tile_sort_long follows k_tile_build_sort (default); 0: the launch is skipped and lists of SORT_SMALL_CAP
                    //    entries and more stay with the sorter below (its global-scratch path) — the trainer's hint, capi.hip
    int mid_sort;   // the same for k_tile_sort_mid and lists of SORT_TINY_CAP entries and more
    int small_first;  // first position of the tile order the short-list sorter's grid covers: short lists in front of it (there
                      //    are few: the order is longest first) are sorted by k_tile_sort_mid.  0: its grid covers the whole order
    int mid_grid;     // workgroups per camera of k_tile_sort_mid (0: a default); it walks on in strides when the head is longer
    int epoch;        // 1 .. 255: the mark of this launch's gradient rows in Scratch::row_epoch; 0: no row marks (every entry owns a row)
    int cut;          // 1: this step's tile lists hold only the entries up to the depth bound the previous step's forward left per tile
                      //    (Scratch::tile_zcut) — the entries behind it were never looked at by any pixel of the tile then; the forward
                      //    checks that they still are not (flags[g * 4 + 0] bit 1 = the cut was wrong: the host replays the step uncut)
    int cut_track;    // 1: this step's forward leaves the bounds for the next step (k_render_fwd<true>); a cut step always tracks
    int cut_margin;   // entries kept behind the last one the forward looked at when it writes the next bound
    uint32_t marks_min_list;  // with epoch != 0: a camera uses row marks iff its longest tile list (flags[g * 4 + 1], written by the tile
                              // scan of the same step) has at least this many entries — the backward and the per-splat kernel read the same
                              // word, so they agree; 0: always
};


// Device pointers of the scratch.  Arrays marked [G] are per geometry group (camera), [V] per pass.
struct Scratch {
    const gs_view* views;      // [V]  per pass (backgrounds differ inside a group)
    const gs_view* gviews;     // [G]  the camera of each group
    const int* view_group;     // [V]  group of each pass
    const int* group_first;    // [G+1] CSR offsets into group_views
    const int* group_views;    // [V]  passes of each group
    GeomRec* geom;             // [V][Pa]
    uint32_t* tiles_touched;   // [V][Pa]
    uint32_t* point_offsets;   // [V][Pa]  inclusive scan of tiles_touched
    uint32_t* block_sums;      // [G][splat_blocks(Pa)] tiles_touched summed per 256-splat block, then its exclusive prefix
    uint32_t* wg_hist;         // [G][splat_blocks(Pa)][NST] candidates per (256-splat block, super-tile); k_coarse_colscan turns
                               //     every super-tile's column into its exclusive prefix over the blocks
    uint32_t* coarse_count;    // [G][NST] candidates per super-tile
    uint32_t* colscan_partial; // [G][chunks][NST] chunk sums of the two-pass column scan (large models only; null: one-pass form)
    uint32_t* coarse_end;      // [G][NST] inclusive scan of coarse_count
    uint4* coarse_list;        // [V][Rcap] {splat id, rect_min, rect_max, first slot} per (splat, super-tile)
    uint32_t* coarse_depth;    // [V][Rcap] depth bits of the same entries
    uint32_t* tile_count;      // [V][T]
    uint32_t* tile_end;        // [V][T]   inclusive scan of tile_count
    uint32_t* tile_order;      // [V][T]   tiles by descending entry count: the order workgroups take them in
    uint32_t* tile_zcut;       // [G][T]   depth bits: every pixel of the tile finished in front of this depth in the newest forward (0xFFFFFFFF: no
                               //          bound — some pixel looked at the whole list); trainer only, else null.  Written by k_render_fwd,
                               //          read by the NEXT step's k_tile_count / k_tile_scatter when Dims::cut
    uint32_t* stile_zcut;      // [G][NST] the largest bound of a super-tile's tiles (k_stile_zcut, first launch of a cut step): a candidate behind
                               //          it is behind every tile's bound — the projection does not count it and the coarse scatter does not emit it
    uint32_t* sort_marks;      // [G][2]   lengths of the order's heads that hold every list of SORT_SMALL_CAP / SORT_TINY_CAP entries and more
    uint32_t* id_of_slot;      // [V][Rcap] (only tiles longer than the rank-sort limit use it)
    uint32_t* point_list;      // [V][Rcap]  sorted splat ids
    uint32_t* slot_list;       // [V][Rcap]  sorted slots
    float* G;                  // [V][Rcap][G_STRIDE]
    uint8_t* row_epoch;        // [V][Rcap]  a gradient row exists this step iff its byte equals Dims::epoch: the backward writes rows (and
                               //            marks them) only for entries some pixel block evaluated; everything else is an implicit zero row
                               //            that is neither written nor read (a dense scene: 84 % of all rows)
    unsigned long long* hit_masks;  // [G][hit_mask_words(Rcap, T)][4]  per (tile, 64-entry sub-block, wave): which entries can reach the
                               // wave's 8x8 block — the forward's ballots, reused by the backward (null: the backward tests again)
    const float* zero_row;     // sixteen zero floats in front of row_epoch's allocation: what k_splat_bwd_view reads in place of an absent row
    float* splat_grads;        // [V][Pa][16]  per-(view,splat) backward records (trainer only)
    float* sh_jac;             // [G][Pa][12]  d colour / d view direction (9 used), written by the projection (trainer only, else null)
    float* mean_copy;          // [3][Pa]  the positions this step's projection read (trainer only, else null): the fused update's SH parts form their
                               //          view directions from it while the geometry part of the same splat may already have moved the position plane
    const uint16_t* sh16;      // [3M][Pa] IEEE half read copy of the SH planes (trainer option "sh_fp16"), or null: read the fp32 planes
    float* out_color;          // [V][3][N]
    float* final_T;            // [V][N]
    uint32_t* n_contrib;       // [V][N]
    const uint32_t* truth;     // [V][N] or null
    const float* dL_dpix;      // [V][3][N] or null (then loss = truth/255 - colour is fused)
    uint32_t* flags;           // [V][4]: 0 bit 0 arena overflow, bit 1 the depth cut of this step was wrong (k_render_fwd); 1 max tile list; 2 entries; 3 sort hints
    float* loss;               // [V][T] per-tile sum of residual^2 (only when truth != null); launch_loss_sum folds it
    float* loss_total;         // [V]
};

// Does geometry group g use row marks in this launch?  (k_render.hip writes them, k_splat_bwd.hip reads them.)
__device__ inline bool uses_row_marks(const Dims& d, const Scratch& s, int g) { return d.epoch != 0 && s.flags[g * 4 + 1] >= d.marks_min_list; }

// ---------------------------------------------------------------------------------------------
// kernel launchers (one translation unit each)
// ---------------------------------------------------------------------------------------------
int launch_preprocess(const Dims& d, const float* params, const Scratch& s, hipStream_t st);
int launch_stile_zcut(const Dims& d, const Scratch& s, hipStream_t st);   // Dims::cut: per super-tile, the largest depth bound of its tiles
// after preprocess, one launch: column scan of the (block x super-tile) count matrix + the prefix of the per-block tile
// sums (-> flags: num_rendered, arena overflow); the scan of the super-tile totals happens inside k_coarse_scatter
int launch_coarse_colscan(const Dims& d, const Scratch& s, hipStream_t st);
size_t colscan_partial_words(int Pa, int NST, int V);  // scratch the two-pass column scan needs (0: one-pass form)
// batched inclusive scan of u32: one workgroup per batch entry up to g_scan_single_max items, three phases beyond
int launch_scan_u32(const uint32_t* in, uint32_t* out, int n, int stride, int batch, uint32_t* partials, hipStream_t st);
size_t scan_partials_count(int n, int batch);
int allow_dynamic_lds(const void* kernel, size_t bytes);  // hipFuncSetAttribute for dynamic LDS beyond the default
extern int g_scan_single_max;
__host__ __device__ inline int splat_blocks(int Pa) { return (Pa + WG - 1) / WG; }
int launch_coarse_scatter(const Dims& d, const Scratch& s, hipStream_t st);
int launch_tile_count(const Dims& d, const Scratch& s, hipStream_t st);
// per-tile segments filled from the super-tile lists + tile_end + longest-first tile order (one launch)
int launch_tile_scatter(const Dims& d, const Scratch& s, uint32_t* partials, hipStream_t st);
int launch_tile_build_sort(const Dims& d, const Scratch& s, hipStream_t st);
int launch_render_forward(const Dims& d, const Scratch& s, hipStream_t st);
// loss_total[v] = sum over tiles of loss[v][tile], in a fixed order (the statistic is reproducible bit for bit)
int launch_loss_sum(const Dims& d, const Scratch& s, hipStream_t st);
int launch_render_backward(const Dims& d, const Scratch& s, const int* items, int n_pairs, int n_singles, bool fuse_pairs, hipStream_t st);
// Trainer form: loops the views, writes the averaged-gradient planes (incl. var) once.  fuse_pairs: the backward left ONE
// gradient set per pair item (render_bwd<2>): the per-splat chain runs once per item and `var` is written as zero.
// Buffers of the compact data-parallel exchange (gs_trainer_set_compact_exchange): what leaves the rank instead of the 12 + 3M
// gradient planes.  48 of cfg3's 60 planes are SH gradients, and a record's SH gradient of a splat is rank one — basis(view
// direction)[M] x dL_dRGB[3] — so the ranks all-gather dL_dRGB (3 floats per record and splat) and all-reduce only the
// twelve other planes; every rank rebuilds the SH planes itself (k_sh_rebuild).
struct Exchange {
    float* geo = nullptr;  // [12][Pa]  loc 3 | scale 3 | opacity 1 | rot 4 | var 1: this rank's sums -> (all-reduce) -> the iteration's
    float* rgb = nullptr;  // [world][hdr + slots * 3 * Pa]  per rank: the positions of the rank's cameras (3 floats per local camera, zero-padded to
                           // hdr floats), then dL_dRGB of every record [slots][3][Pa]; this rank writes chunk `rank` -> (all-gather).  The positions
                           // travel WITH the records they belong to: k_sh_rebuild forms every record's view direction from the gathered header, so
                           // cameras that moved since the exchange was installed (a re-capture, gs_trainer_set_views) cannot leave a stale basis behind
    int rank = 0, world = 1;
    int slots = 0;         // records per rank: ceil(cameras / world) in the fused form, twice that (white | black) in the per-pass form
    int hdr = 0;           // floats in front of a rank's records: 3 * ceil(cameras / world) rounded up to 64 (the planes stay 256-byte aligned)
};
inline int exchange_header_floats(int n_cameras, int world) { return ((3 * ((n_cameras + world - 1) / world)) + 63) / 64 * 64; }
struct FusedUpdate;
int launch_splat_backward_avg(const Dims& d, const float* params, const Scratch& s, float samples, float* grad_planes, const int* items,
                              int n_pairs, int n_singles, bool fuse_pairs, hipStream_t st, const Exchange* x = nullptr,
                              const FusedUpdate* fu = nullptr, bool* update_applied = nullptr);
// after the exchange: every averaged-gradient plane from the reduced geometry planes and the gathered dL_dRGB + camera positions
// parts: 1 = the SH planes (from the gathered records), 2 = the twelve other planes (from the reduced sums), 3 = both
// fu (optional): the update fused into the rebuild (every part updates the planes it has just written); mean_copy: the projection's copy of the positions
int launch_sh_rebuild(const Dims& d, const float* params, const float* mean_copy, const Exchange& x, int n_cameras, bool per_pass, float samples,
                      float* grad_planes, int parts, hipStream_t st, const FusedUpdate* fu = nullptr);
// Seam form: one view, reference-shaped AoS outputs with the reference's += / = discipline.
struct SeamGrads { float *dL_dmean2D, *dL_dconic, *dL_dopacity, *dL_dcolor, *dL_dmean3D, *dL_dcov3D, *dL_dsh, *dL_dscale, *dL_drot; };
int launch_splat_backward_seam(const Dims& d, const float* params, const Scratch& s, const SeamGrads& g, hipStream_t st);
// applyGradients (src/Trainer.cu:81-101) / the Adam extension for ONE element: the same fp32 operations wherever they run — the
// update launch (k_update.hip) and the fused form inside the per-splat reduction (k_splat_bwd.hip, k_splat_bwd_reduce<D, true>).
struct UpdateArgs {
    float lr_loc, lr_sh, lr_scale, lr_opac, lr_rot, scale_max;
    int rule;
    float b1, b2, eps, bc1, bc2;
    int M;
};
UpdateArgs make_update_args(const Planes& pl, const gs_hyper& h, int adam_t);
#ifdef __HIPCC__
// kind: 0 plain, 1 scale clamp, 2 opacity clamp
__device__ inline void update_plane_rule(const UpdateArgs& u, const Planes& pl, int p, float& lr, int& kind) {
    if (p < 3) { lr = u.lr_loc; kind = 0; }
    else if (p < pl.scale(0)) { lr = u.lr_sh; kind = 0; }
    else if (p < pl.opac()) { lr = u.lr_scale; kind = 1; }
    else if (p == pl.opac()) { lr = u.lr_opac; kind = 2; }
    else { lr = u.lr_rot; kind = 0; }
}
// the rule on values: returns the new parameter, moves the moments (Adam only)
__device__ inline float update_value(const UpdateArgs& u, float lr, int kind, float x, float g, float& m, float& v) {
    if (u.rule == GS_UPDATE_ADAM) {
        m = u.b1 * m + (1.0f - u.b1) * g;
        v = u.b2 * v + (1.0f - u.b2) * g * g;
        const float mh = m / u.bc1, vh = v / u.bc2;
        x = x + lr * (mh / (sqrtf(vh) + u.eps));
    } else {
        x = x + g * lr;
    }
    if (kind == 1) x = fminf(u.scale_max, fmaxf(0.0f, x));
    else if (kind == 2) x = fminf(1.0f, fmaxf(0.0f, x));
    return x;
}
__device__ inline float update_element(const UpdateArgs& u, float lr, int kind, float x, float g, float* am, float* av, size_t idx) {
    float m = 0.0f, v = 0.0f;
    if (u.rule == GS_UPDATE_ADAM) { m = am[idx]; v = av[idx]; }
    x = update_value(u, lr, kind, x, g, m, v);
    if (u.rule == GS_UPDATE_ADAM) { am[idx] = m; av[idx] = v; }
    return x;
}
#endif
// The update fused into the per-splat reduction of a step that has no collective between the two (north_star: "fused Adam update"):
// the thread that has just formed a splat's averaged gradients applies them — the gradient planes are still stored (tests,
// gs_trainer_grad_buffer, densify read them) but not read back, and the update launch is gone.  The arena-overflow verdict of the
// step is on the device by then (Scratch::flags): an attempt that overflowed applies nothing, the host replays it.
struct FusedUpdate {
    UpdateArgs u;
    float* params = nullptr;   // null: not fused
    float *am = nullptr, *av = nullptr;
    uint16_t* sh16 = nullptr;
};
// updates the flat element range [lo, hi) of the parameter planes (default: all of them)
// sh16 != null: the fp16 read copy of every SH element this launch updates is refreshed in the same pass
int launch_update(const Planes& pl, int P, int Pa, float* params, float* grads, float* adam_m, float* adam_v, int adam_t,
                  const gs_hyper& h, hipStream_t st, size_t lo = 0, size_t hi = ~(size_t)0, uint16_t* sh16 = nullptr);
// fp16 read copy of all SH planes: sh16[k * Pa + i] = half(planes[(3 + k) * Pa + i]), round to nearest even
int launch_sh_to_half(int M, int P, int Pa, const float* planes, uint16_t* sh16, hipStream_t st);
// Floats every plane-major buffer of a model is allocated with: the 11+3M parameter planes, one spare plane (the
// gradient buffer's `var`), and padding so that the buffer divides into equal chunks for any rank count <= 64 (the
// sharded update reduce-scatters gradients and all-gathers parameters with ONE chunking, gs_trainer_set_sharded_update).
// 64-entry sub-blocks of all tile lists of a group, indexed (tile_start >> 6) + tile + sub_block (lists are not 64-aligned:
// consecutive tiles never collide under this index)
__host__ __device__ inline size_t hit_mask_words(uint32_t Rcap, int T) { return ((size_t)Rcap >> 6) + (size_t)T + 1; }
inline size_t plane_buffer_floats(int M, int Pa) { return (size_t)(12 + 3 * M) * Pa + 64 * 64; }
inline size_t shard_total_floats(int M, int Pa, int world) {
    const size_t n = (size_t)(12 + 3 * M) * Pa, q = (size_t)world * 64;
    return (n + q - 1) / q * q;
}
int launch_aos_to_soa(int P, int Pa, int M, const float* loc, const float* sh, const float* scale, const float* opac,
                      const float* rot, float* planes, hipStream_t st);
int launch_soa_to_aos(int P, int Pa, int M, const float* planes, float* loc, float* sh, float* scale, float* opac,
                      float* rot, hipStream_t st);
int launch_copy_probe(const void* src, void* dst, size_t bytes, int form, hipStream_t st);   // form 0..15: grid size, unroll, non-temporal
int launch_image_float_to_int(const float* src, uint32_t* fb, int w, int h, hipStream_t st);
int launch_image_int_to_loss(const uint32_t* truth, const float* rast, float* loss, int w, int h, hipStream_t st);
int launch_ranges(const Dims& d, const Scratch& s, uint32_t* ranges, hipStream_t st);
int launch_debug_reduce9(const float* in, float* out, hipStream_t st);
int debug_counters(unsigned long long out[8], bool reset);  // k_render.hip: the GS_DIAG_COUNT_ACTIVE build's lane counters

// densify / prune on the device (k_densify.hip; the reference does it on the CPU, src/Trainer.cu:433-542)
int launch_densify_classify(int count, int Pa, int M, const float* params, const float* grad, const gs_hyper& h, uint32_t* flags,
                            uint32_t* ranks, int fs, uint32_t* partials, hipStream_t st);
int launch_densify_emit(int count, int Pa, int M, const float* params, const float* grad, const gs_hyper& h, const uint32_t* flags,
                        const uint32_t* ranks, int fs, int splits_done, int clones_done, int kept, int outPa, float* out, hipStream_t st);

// Adam moments (two [11+3M][Pa] buffers) re-indexed like the parameters of the same densify (k_densify.hip)
int launch_densify_carry(int count, int Pa, int M, const gs_hyper& h, const uint32_t* flags, const uint32_t* ranks, int fs, int splits_done,
                         int clones_done, int kept, int outPa, const float* src_m, float* dst_m, const float* src_v, float* dst_v,
                         hipStream_t st);

}  // namespace gs

struct gs_model {
    int capacity = 0, sh_degree = 0, sh_coeffs = 0, count = 0;
    int device = 0;
    int Pa = 0;            // plane stride of `planes`
    float* planes = nullptr;  // [11+3M][Pa] device
    size_t planes_bytes = 0;  // size of that allocation (densify swaps plane sets with the trainer's spare one: it may be larger than Pa needs)
};
