// k_render.hip — forward alpha-compositing raster and per-pixel backward pass.
// Replaces renderCUDA (forward) and renderCUDA (backward) inside CudaRasterizer::Rasterizer::forward /
// ::backward (reference call sites src/Trainer.cu:334-360, :378-412)
// and fuses the reference's imageIntToLoss (src/Trainer.cu:33-44) into the backward prologue.
//
// MI355X mapping (not the upstream one):
//   * workgroup = one 16x16 tile = 4 wave64; each WAVE owns an 8x8 pixel block, lane = pixel.
//   * the tile's depth-ordered list is staged through LDS (48-byte records gathered from the
//     64-byte per-splat lines written by preprocess, conic pre-scaled to base 2): 256 entries per round
//     forward, 64 backward (12.5 / 20.9 KB of LDS per workgroup, 8 / 7 waves per SIMD).
//   * workgroups take tiles longest list first (tile_order), so a launch ends on light tiles.
//   * per 64 staged entries every lane tests ONE entry's "alpha >= 1/255" box against the wave's
//     8x8 block; the ballot is a scalar bit list and only surviving entries are evaluated — the
//     skipped pairs are exactly ones the reference blend would skip too, so results are unchanged.
//     (Reading the next surviving entry's record one iteration early was measured slower: at 8 waves
//     per SIMD the LDS latency is covered and the extra selects cost issue slots.)
//   * backward: a hit parks two values per pixel (w = alpha T, u = G dL/dalpha) in a per-wave LDS buffer; every four hits
//     the wave contracts the buffer over the pixels with plain FMAs (lane = (hit, four pixels of a row)) and folds the 16
//     lanes of a hit with a 24-instruction DPP reduce-scatter; each wave keeps its nine sums per entry in its own LDS
//     slot, the four slots are added in fixed order and written as ONE 36-byte row per (splat, tile) entry.  No global
//     atomic anywhere; the result is bitwise reproducible.  A step without densify runs this once per CAMERA on the sum
//     of its passes' residual images (render_bwd<2>), not once per pass.
#include "gs_internal.h"

namespace gs {

constexpr float ALPHA_MIN = 1.0f / 255.0f;
constexpr float ALPHA_MAX = 0.99f;
constexpr float T_STOP = 0.0001f;
constexpr int ACC_STRIDE = 9;
#ifndef GS_BWD_PAIR_WAVES
#define GS_BWD_PAIR_WAVES 7  // occupancy target of the fused-pair backward (tuning hook, tools/build_variant.sh)
#endif

// Exact "can this splat reach alpha >= 1/255 anywhere in the 8x8 block" test: the minimum of the conic's
// quadratic form q(d) = 0.5*(a dx^2 + c dy^2) + b dx dy over the block rectangle (centre inside -> 0, else
// the best of the four edges, each a clamped 1-D parabola) compared with tau = ln(255*opacity) (+margin,
// computed in preprocess; tau < 0: never visible, tau = 3e38: culling disabled / conic not positive definite).
// One lane tests one staged entry, so the ~45 VALU instructions are amortised over 64 entries.
__device__ inline bool block_reaches(float cx, float cy, float a, float b, float c, float tau, float bxlo, float bxhi, float bylo,
                                     float byhi) {
    const float x0 = bxlo - cx, x1 = bxhi - cx, y0 = bylo - cy, y1 = byhi - cy;
    const bool inside = (x0 <= 0.0f) && (x1 >= 0.0f) && (y0 <= 0.0f) && (y1 >= 0.0f);
    const float ra = __builtin_amdgcn_rcpf(a), rc = __builtin_amdgcn_rcpf(c);
    auto q = [&](float dx, float dy) { return 0.5f * (a * dx * dx + c * dy * dy) + b * dx * dy; };
    const float e0 = q(x0, fminf(y1, fmaxf(y0, -b * x0 * rc)));
    const float e1 = q(x1, fminf(y1, fmaxf(y0, -b * x1 * rc)));
    const float e2 = q(fminf(x1, fmaxf(x0, -b * y0 * ra)), y0);
    const float e3 = q(fminf(x1, fmaxf(x0, -b * y1 * ra)), y1);
    const float qmin = inside ? 0.0f : fminf(fminf(e0, e1), fminf(e2, e3));
    return (tau >= 0.0f) && !(qmin > tau);
}

// Staged entries carry the conic pre-scaled for the blend loop: the exponent is evaluated directly in base 2,
//   log2(G) = dx * (ha*dx + nb*dy) + hc*dy*dy,   ha = -0.5*log2e*conic.x, nb = -log2e*conic.y, hc = -0.5*log2e*conic.z,
// five VALU instructions instead of seven plus the log2e multiply in front of v_exp_f32; the scaling is done once
// per staged entry instead of once per (entry, pixel) pair.  tau is scaled by log2e as well.
constexpr float LOG2E = 1.4426950408889634f;
template <int N> struct StagedTile {  // three 16-byte-strided arrays: one scalar address serves all reads of an entry
    float4 A[N];  // x, y, ha, nb
    float4 B[N];  // hc, opacity, r, g
    float4 C[N];  // b, tau*log2e (cull threshold), -, depth
};

template <int N> __device__ inline void stage_entry(StagedTile<N>& t, int slot, const GeomRec* __restrict__ r) {
    const float4* q = reinterpret_cast<const float4*>(r);
    float4 a = q[0], b = q[1], c = q[2];
    a.z *= -0.5f * LOG2E; a.w *= -LOG2E; b.x *= -0.5f * LOG2E; c.y *= LOG2E;
    t.A[slot] = a; t.B[slot] = b; t.C[slot] = c;
}
// the conic back from a staged entry (flush of the backward kernel); exact up to one rounding
__device__ inline void unscaled_conic(const float4& A, const float4& B, float& conA, float& conB, float& conC) {
    conA = A.z * (-2.0f / LOG2E); conB = A.w * (-1.0f / LOG2E); conC = B.x * (-2.0f / LOG2E);
}

// Diagnostic counters (gs_debug_counters): with -DGS_DIAG_COUNT_ACTIVE (tools/build_variant.sh) the two blend loops count,
// per evaluated (entry, 8x8 block) pair — a "hit" —, how many of the wave's 64 lanes do useful work: the pixel is still
// blending and the pair passes the alpha >= 1/255 test.  The hit is branch-free, so EXEC-based hardware counters cannot see
// this.  [0] forward hits, [1] forward active lanes, [2] backward hits, [3] backward active lanes, [4] / [5] forward / backward
// staged (entry, block) pairs before the block test (what a kernel without the exact block test would evaluate); [6] / [7]
// backward only: the iterations the same rounds would take if hits were packed by 8x4 half / by 4x4 quadrant (sum over
// rounds of the largest per-half / per-quadrant hit count) — the ceiling of any finer-grained scheme.
__device__ unsigned long long g_diag_counters[8];
int debug_counters(unsigned long long out[8], bool reset) {
    GS_HIP(hipDeviceSynchronize());
    GS_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_diag_counters), 8 * sizeof(unsigned long long)));
    if (reset) { const unsigned long long z[8] = { 0 }; GS_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_diag_counters), z, sizeof(z))); }
    return GS_OK;
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
// TRACK (a step whose NEXT step may cut its lists, Dims::cut_track): the forward also leaves every tile's depth bound and checks this step's cut
template <bool TRACK>
__global__ __launch_bounds__(WG) void k_render_fwd(Dims d, Scratch s) {
#pragma clang fp contract(fast)
    __shared__ StagedTile<WG> st;
    const int v = blockIdx.y;  // v = geometry group (camera)
    if (s.flags[v * 4 + 0] & 1u) return;  // overflowed arena: the binning skipped the group, tile_order / lists hold nothing of this step
#ifdef GS_DIAG_XCD_STRIPES  // timing experiment: workgroups b, b+8, b+16 ... (one XCD) take a contiguous stripe of tiles, row-major
    const int tile = (int)((blockIdx.x % 8) * (d.T / 8) + blockIdx.x / 8);
#else
    const int tile = (int)s.tile_order[(size_t)v * d.T + blockIdx.x];
#endif
    if ((unsigned)tile >= (unsigned)d.T) return;  // an order that is not a permutation of the tiles must never become an address (round-3 fault)
    const int tx = tile % d.gx, ty = tile / d.gx;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int bx0 = tx * TILE + (wave & 1) * 8, by0 = ty * TILE + (wave >> 1) * 8;
    const int px = bx0 + (lane & 7), py = by0 + (lane >> 3);
    const bool inside = px < d.W && py < d.H;
    const float pxf = (float)px, pyf = (float)py;
    const float bxlo = (float)bx0, bxhi = (float)(bx0 + 7), bylo = (float)by0, byhi = (float)(by0 + 7);

    const int n = (int)s.tile_count[(size_t)v * d.T + tile];
    const uint32_t start = s.tile_end[(size_t)v * d.T + tile] - (uint32_t)n;
    const uint32_t* __restrict__ plist = s.point_list + (size_t)v * d.Rcap + start;
    const GeomRec* __restrict__ geom = s.geom + (size_t)v * d.Pa;
    unsigned long long* const hm = s.hit_masks ? s.hit_masks + ((size_t)v * hit_mask_words(d.Rcap, d.T) + (start >> 6) + (uint32_t)tile) * 4 : nullptr;

    // Branch-free per-lane blend.  Tw > 0 is the transmittance of a pixel that still blends; once the pixel saturates
    // (upstream's `done`) or for pixels outside the image Tw <= 0 holds MINUS the transmittance it stopped at, which
    // makes every later product fail the T_STOP test by itself — no separate `live` flag, no select per state
    // variable (v_cndmask costs 4.4 SIMD-cycles on gfx950 against 2.7 for an fma).  The only control flow in the pair
    // loop is the scalar walk over the ballot bits.
    float Tw = inside ? 1.0f : -1.0f, C0 = 0.0f, C1 = 0.0f, C2 = 0.0f;
    uint32_t last = 0;
    const uint32_t bound_used = (TRACK && d.cut && tid == 0) ? s.tile_zcut[(size_t)v * d.T + tile] : 0xFFFFFFFFu;   // this step's list was built under it
    int walked = 0;   // list positions this wave looked at (uniform per wave): what the next step's depth cut must keep
#ifdef GS_DIAG_COUNT_ACTIVE
    unsigned long long diag_hits = 0, diag_active = 0, diag_staged = 0;
#endif

    for (int base = 0; base < n; base += WG) {
        if (__syncthreads_and(!(Tw > 0.0f))) break;  // whole tile saturated; also guards LDS reuse
        const int e = base + tid;
        if (e < n) stage_entry(st, tid, geom + plist[e]);
        __syncthreads();
        const int cnt = min(WG, n - base);
        for (int sub = 0; sub < cnt; sub += 64) {
            if (__ballot(Tw > 0.0f) == 0ull) break;
            if constexpr (TRACK) walked = base + min(sub + 64, cnt);
            const int j = sub + lane;
            bool hit = false;
            if (j < cnt) {
                const float4 a = st.A[j];
                hit = block_reaches(a.x, a.y, -2.0f * a.z, -a.w, -2.0f * st.B[j].x, st.C[j].y, bxlo, bxhi, bylo, byhi);
            }
            unsigned long long mask = __ballot(hit);
#ifdef GS_DIAG_COUNT_ACTIVE
            diag_staged += (unsigned long long)min(64, cnt - sub);
#endif
            // the backward of the same camera walks the same lists against the same blocks: it reuses this ballot
            if (hm && lane == 0) hm[(size_t)((base + sub) >> 6) * 4 + wave] = mask;
            // The pixel's last contributor is recorded as the entry's BYTE OFFSET in the staging arrays — the register that already
            // holds the LDS address of the three reads — and turned into a list position once per 64 entries: a position formed on
            // the scalar unit would cost a v_mov per hit to reach the select (v_cndmask takes one scalar operand, the mask).
            uint32_t last_off = 0xffffffffu;
            while (mask) {
                const int k = __builtin_ctzll(mask);
                asm("s_bitset0_b64 %0, %1" : "+s"(mask) : "s"(k));  // one SALU instruction instead of the add / addc / and of mask &= mask - 1
                const uint32_t off = (uint32_t)(sub + k) * 16u;
                const char* const rec = reinterpret_cast<const char*>(&st) + off;
                const float4 A = *reinterpret_cast<const float4*>(rec), B = *reinterpret_cast<const float4*>(rec + sizeof(st.A));
                const float cb = *reinterpret_cast<const float*>(rec + sizeof(st.A) + sizeof(st.B));
                const float dx = A.x - pxf, dy = A.y - pyf;
                const float power = dx * (A.z * dx + A.w * dy) + B.x * dy * dy;  // log2 of the Gaussian weight
                float alpha = fminf(ALPHA_MAX, B.y * __builtin_amdgcn_exp2f(power));
                const bool valid = (power <= 0.0f) & (alpha >= ALPHA_MIN);
                alpha = valid ? alpha : 0.0f;
                const float test_T = Tw * (1.0f - alpha);
                const bool go = test_T >= T_STOP;  // false for finished pixels and for the entry that finishes one (not applied)
                const float w = alpha * (go ? Tw : 0.0f);
                Tw = go ? test_T : -fabsf(Tw);
                C0 += B.z * w; C1 += B.w * w; C2 += cb * w;
                last_off = (go & valid) ? off : last_off;
#ifdef GS_DIAG_COUNT_ACTIVE
                diag_hits++; diag_active += (unsigned long long)__popcll(__ballot(go & valid));
#endif
            }
            last = last_off != 0xffffffffu ? (uint32_t)base + (last_off >> 4) + 1u : last;
        }
    }
#ifdef GS_DIAG_COUNT_ACTIVE
    if (lane == 0) { atomicAdd(&g_diag_counters[0], diag_hits); atomicAdd(&g_diag_counters[1], diag_active); atomicAdd(&g_diag_counters[4], diag_staged); }
#endif
    if constexpr (TRACK) {
        // The depth bound for the NEXT step's lists (Dims::cut, k_tile_count / k_tile_scatter): if every pixel of the tile finished
        // (T below 1e-4, or outside the image), nothing behind the last position a wave looked at was read — the bound is the depth of
        // the entry cut_margin positions further on (the model moves a little between steps); else there is no bound.  And the check
        // of THIS step's cut: a pixel that is still blending at the end of a list the cut shortened would have gone on into the
        // dropped entries — the step is wrong from here on, everything behind the forward skips the camera and the host replays it.
        // (A tile with a finite bound counts as shortened whether or not an entry actually lay behind the bound.)
        __shared__ int s_walked;
        __syncthreads();     // (the loop's last LDS reads are done: s_walked may share the staging area's bank, not its bytes)
        if (tid == 0) s_walked = 0;
        const int any_alive = __syncthreads_or(Tw > 0.0f);
        if (lane == 0) atomicMax(&s_walked, walked);
        __syncthreads();
        if (tid == 0) {
            // (a list that is already cut and was needed nearly to its end keeps its bound: asking for the depth of a position behind
            //  the cut list would mean "no bound", and the tile would alternate between cut and uncut steps)
            uint32_t bound = 0xFFFFFFFFu;
            const int keep = s_walked + d.cut_margin;
            //  — unless the pixels have come within a quarter of the margin of the cut list's end: then the tile asks for a full list once
            //  ("no bound") and takes a fresh bound from it, instead of waiting for the cut to be found wrong, which costs the whole step)
            if (!any_alive && keep > 0)
                bound = keep < n ? __float_as_uint(geom[plist[keep - 1]].depth) : (s_walked + d.cut_margin / 4 < n ? bound_used : 0xFFFFFFFFu);
            s.tile_zcut[(size_t)v * d.T + tile] = bound;
            if (d.cut && any_alive && bound_used != 0xFFFFFFFFu) atomicOr(&s.flags[v * 4 + 0], 2u);
        }
    }
    const float T = fabsf(Tw);
    if (inside) {
        const size_t pix = (size_t)py * d.W + px;
        s.final_T[(size_t)v * d.N + pix] = T;
        s.n_contrib[(size_t)v * d.N + pix] = last;
        // every pass of this camera gets its own image: same blend, its own background
        for (int k = s.group_first[v]; k < s.group_first[v + 1]; k++) {
            const int u = s.group_views[k];
            const float* bg = s.views[u].bg;
            float* out = s.out_color + (size_t)u * 3 * d.N;
            out[pix] = C0 + T * bg[0];
            out[(size_t)d.N + pix] = C1 + T * bg[1];
            out[2 * (size_t)d.N + pix] = C2 + T * bg[2];
        }
    }
}

int launch_render_forward(const Dims& d, const Scratch& s, hipStream_t stream) {
    if (d.T == 0 || d.VG == 0) return GS_OK;
    if (d.cut_track && s.tile_zcut) hipLaunchKernelGGL(k_render_fwd<true>, dim3(d.T, d.VG), dim3(WG), 0, stream, d, s);
    else hipLaunchKernelGGL(k_render_fwd<false>, dim3(d.T, d.VG), dim3(WG), 0, stream, d, s);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

// ---------------------------------------------------------------------------------------------
// wave reductions
// ---------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK = 0xf>
__device__ inline float dpp_add(float x) {
    const int y = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, ROW_MASK, 0xf, false);
    return x + __builtin_bit_cast(float, y);
}
// sum over the 64 lanes; the total is valid in lane 63
__device__ inline float wave_sum_to_lane63(float x) {
    x = dpp_add<0xB1>(x);        // quad_perm [1,0,3,2]
    x = dpp_add<0x4E>(x);        // quad_perm [2,3,0,1]
    x = dpp_add<0x124>(x);       // row_ror:4
    x = dpp_add<0x128>(x);       // row_ror:8
    x = dpp_add<0x142, 0xA>(x);  // row_bcast:15 -> rows 1,3
    x = dpp_add<0x143, 0xC>(x);  // row_bcast:31 -> rows 2,3
    return x;
}

// Reduce-scatter of nine per-lane values over every 16-lane ROW of the wave.  Returns r such that lane 2q (q = 0..7) of
// a row holds the row total of g_q and lane 1 the row total of g8.
//   stage A  lane ^ 8 (row_ror:8): lanes with bit3 = 0 keep g0..g3, lanes with bit3 = 1 keep g4..g7
//   stage B  bank ^ 1 (row_ror:12 / row_ror:4): bit2 selects the lower / upper two of the four
//   stage C  lane ^ 2 (quad_perm):               bit1 selects one of the two
//   stage D  lane ^ 1
// bank_mask predicates whole 4-lane banks, so stages A and B need no select instructions (24 instructions instead of the
// 36 of nine butterflies).  EXEC must be all ones (the callers are in wave-uniform control flow).
__device__ inline float wave_reduce_scatter9_rows(float g0, float g1, float g2, float g3, float g4, float g5, float g6, float g7,
                                                  float g8) {
    float t0, t1;
    const unsigned long long mask_bit1 = 0xCCCCCCCCCCCCCCCCull;  // lanes with bit 1 set
    const unsigned long long mask_lane1 = 0x0002000200020002ull;  // lane % 16 == 1
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %0, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %1, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %8, %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %2, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %8, %8, %8 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %3, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %8, %8, %8 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 row_ror:12 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %0, %2, %2 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %8, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_ror:12 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %1, %3, %3 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_cndmask_b32_e64 %9, %0, %1, %11\n\t"
        "v_cndmask_b32_e64 %10, %1, %0, %11\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %9, %10, %9 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %9, %9, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_e64 %9, %9, %8, %12\n\t"
        : "+v"(g0), "+v"(g1), "+v"(g2), "+v"(g3), "+v"(g4), "+v"(g5), "+v"(g6), "+v"(g7), "+v"(g8), "=&v"(t0), "=&v"(t1)
        : "s"(mask_bit1), "s"(mask_lane1));
    return t0;
}

// debug entry: one wave, in[q][lane] -> out[lane] = wave_reduce_scatter9_rows(...) (tests/test_gpu_raster.py)
__global__ void k_debug_reduce9(const float* __restrict__ in, float* __restrict__ out) {
    const int l = threadIdx.x;
    out[l] = wave_reduce_scatter9_rows(in[l], in[64 + l], in[128 + l], in[192 + l], in[256 + l], in[320 + l], in[384 + l],
                                  in[448 + l], in[512 + l]);
}
int launch_debug_reduce9(const float* in, float* out, hipStream_t st) {
    hipLaunchKernelGGL(k_debug_reduce9, dim3(1), dim3(64), 0, st, in, out);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

// ---------------------------------------------------------------------------------------------
// pixel sums of the backward by LDS hand-off
// ---------------------------------------------------------------------------------------------
// The nine per-(entry, 8x8 block) sums are contractions over the block's 64 pixels of two per-pixel values,
//   w = alpha * T (with the per-pixel constant dL/dpixel)   and   u = G * dL/dalpha (with 1, dx, dy, dx^2, dx dy, dy^2).
// A hit therefore parks only (w, u) of its 64 lanes in a per-wave LDS buffer (one ds_write2st64_b32) and goes on; after
// PARK_BATCH = 4 hits the wave contracts the buffer with PLAIN fp32 FMAs: lane (hit h = lane / 16, sub = lane % 16) reads
// pixels 4 sub .. 4 sub + 3 of hit h (four consecutive pixels of one row; conflict-free ds_read_b64: the hit stride of 520
// bytes puts the 32 lanes of a half wave on 32 different 8-byte bank pairs), forms the partial colour sums and the
// moments of u in dx — dy is constant for the lane — and the 16 lanes of a hit, one DPP row, are folded
// by the 24-instruction reduce-scatter above.  Per hit: ~8 full-rate VALU + 6 DPP, against the 9 multiplies + 24
// half-rate DPP adds + 2 ds_bpermute of reducing every hit in registers (round 2: 0.90 ms per 16-view launch; this
// form 0.66).  A batch of 8 hits (8 lanes x 8 pixels, a 20-instruction fold) was built as well: its 16.6 KB of parking
// space and 24 registers of dL/dpixel limit the kernel to 5 waves per SIMD, where it is latency-bound (0.90 ms).
constexpr int PARK_BATCH = 4;             // hits per contraction
constexpr int PARK_LPH = 64 / PARK_BATCH; // lanes per hit: one 16-lane DPP row
constexpr int PARK_PPL = 64 / PARK_LPH;   // pixels per lane: four consecutive pixels of one row of the 8x8 block
constexpr int PARK_STRIDE = 2 * 64 + 2;   // floats per parked hit: w[64] | u[64] | 8 bytes of padding (conflict-free ds_read_b64)

// Contracts the wave's parked hits (nb <= PARK_BATCH of them; their staging slots packed 8 bits apiece in blo, newest in
// the low byte) and stores the nine sums of every hit in the wave's accumulator slots acc[slot * ACC_STRIDE + q].
//   dpr[c][i]  dL/dpixel channel c of this lane's i-th pixel;  pxcol0 = x of its first pixel;  pyrow = y of its row.
template <int N>
__device__ __forceinline__ void contract_parked(const float* __restrict__ park, const StagedTile<N>& st, float* __restrict__ acc, int nb,
                                                uint32_t blo, const float (&dpr)[3][PARK_PPL], float pxcol0, float pyrow, int lane) {
    const int h = lane / PARK_LPH, sub = lane % PARK_LPH;
    const int jj = (int)__builtin_amdgcn_ubfe(blo, (uint32_t)(8 * (nb - 1 - h)) & 31u, 8u);  // hit h of the batch (parking order)
    const float2 xy = *reinterpret_cast<const float2*>(&st.A[jj]);
    const float2* row = reinterpret_cast<const float2*>(park + h * PARK_STRIDE + sub * PARK_PPL);
    float w[PARK_PPL], u[PARK_PPL];
#pragma unroll
    for (int k = 0; k < PARK_PPL / 2; k++) { const float2 t = row[k]; w[2 * k] = t.x; w[2 * k + 1] = t.y; }
#pragma unroll
    for (int k = 0; k < PARK_PPL / 2; k++) { const float2 t = row[32 + k]; u[2 * k] = t.x; u[2 * k + 1] = t.y; }
    // colour sums and the moments of u; dx is formed per pixel exactly as the hit formed it.  (Moments over the pixel INDEX,
    // sum u dx^2 = X0 (X0 n0 - 2 n1) + n2 with X0 = x - first pixel, cost 7 VALU less per contraction and were built: they
    // cancel catastrophically for a splat centred on a pixel column whose neighbours fall under the alpha cut — one live
    // pixel with dx ~ 0.01 per lane — and failed the long-list parity test by 15 x sum|term|.)
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, m0 = 0.0f, m1 = 0.0f, m2 = 0.0f;
#pragma unroll
    for (int i = 0; i < PARK_PPL; i++) {
        const float dx = xy.x - (pxcol0 + (float)i);
        const float ux = u[i] * dx;
        m0 += u[i]; m1 += ux; m2 = fmaf(ux, dx, m2);
        a0 = fmaf(w[i], dpr[0][i], a0); a1 = fmaf(w[i], dpr[1][i], a1); a2 = fmaf(w[i], dpr[2][i], a2);
    }
    const float dy = xy.y - pyrow;
    const float m0y = m0 * dy;
    // order of the nine sums as everywhere else: colour(3), u dx, u dy, u dx dx, u dx dy, u dy dy, u
    const float red = wave_reduce_scatter9_rows(a0, a1, a2, m1, m0y, m2, m1 * dy, m0y * dy, m0);  // per 16-lane row: lane 2q -> q, lane 1 -> 8
    if (h < nb && ((sub & 1) == 0 || sub == 1)) acc[jj * ACC_STRIDE + ((sub & 1) ? 8 : (sub >> 1))] = red;
}

// ---------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------
// One workgroup = one tile of one camera and ONE gradient set: the sum of F passes' residual images.
//   <F = 1>  a single pass: the rasterizer seam, and every pass of a step that needs per-pass gradients —
//            accumulateGradients' `var += |g_loc| / S` (src/Trainer.cu:52) takes the norm of every pass's location
//            gradient (densify steps, gs_trainer_accumulate).
//   <F = 2>  the two passes of a camera (the reference's white / black pair) at once.  Every other output of
//            accumulateGradients is a plain sum over the passes, and the whole backward is linear in dL/dpixel for a fixed
//            camera, so the two residual images are added per pixel (dL/dpixel = r_white + r_black, and
//            T_final * (bg . r) likewise) and the pair costs one single-pass backward.  `var` is not produced — the trainer
//            takes this form on steps whose `var` nobody reads (gs_trainer_step without densify; the reference recomputes
//            var from zero every iteration and reads it only inside the densify block, src/Trainer.cu:304-309,444,453).
// Structure of a round (64 entries):
//   * ONE ballot: lane l of every wave tests slot l against the wave's 8x8 block; slot s of the staging area holds entry
//     63 - s of the round, so walking the ballot's bits upwards (s_ff1 + s_bitset0) is the back-to-front traversal;
//   * the hit is branch-free: an inactive lane (behind the pixel's last contributor, power > 0, alpha < 1/255) continues
//     with G = alpha = 0, which leaves T and the accumulated colour unchanged and parks exact zeros — two v_cndmask
//     instead of an exec-masked region, and every hit is parked (no "did any lane act" test: the touched bits of the
//     round are the ballot itself);
//   * all loop state (hit counter, packed slot indices, ballot) is wave-uniform and lives in SGPRs.
// The scalar unit is shared by the CU's four SIMDs (4.7 cycles per SALU instruction per SIMD, tools/valu_rate.hip): the
// first hand-off version spent 40 SALU instructions per hit on loop control, as much SIMD time as its VALU work.
// (A two-gradient-set form that shared geometry, exp and the transmittance recurrence between the passes of a camera
// existed through round 2; two launches of this kernel take the time it took.)
template <int F>
__device__ __forceinline__ void render_bwd(const Dims& d, const Scratch& s, const int* __restrict__ item) {
#pragma clang fp contract(fast)
    constexpr int ROUND = 64;
    __shared__ StagedTile<ROUND> st;
    __shared__ uint32_t sSlot[ROUND];
    __shared__ float sAcc[4 * ROUND * ACC_STRIDE];
    __shared__ __attribute__((aligned(16))) float sPark[4 * PARK_BATCH * PARK_STRIDE];
    __shared__ unsigned long long sTouched[4];
    __shared__ uint32_t sMaxLast;
    __shared__ float sLoss[F][4];
    const int g = item[0];  // item = {group, pass 0[, pass 1]}; lists, records, T and n_contrib live in the geometry group
#ifdef GS_DIAG_XCD_STRIPES
    const int tile = (int)((blockIdx.x % 8) * (d.T / 8) + blockIdx.x / 8);
#else
    const int tile = (int)s.tile_order[(size_t)g * d.T + blockIdx.x];
#endif
    int vin[F];  // the passes whose residual images are summed; the rows go to the slice of the first one
#pragma unroll
    for (int q = 0; q < F; q++) vin[q] = item[1 + q];
    if (s.flags[g * 4 + 0] & 3u) return;   // arena overflow, or a depth cut the forward found wrong: the host replays the step
    if ((unsigned)tile >= (unsigned)d.T) return;  // see k_render_fwd
    const int tx = tile % d.gx, ty = tile / d.gx;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // uniform: the per-wave roles below are scalar branches
    const int bx0 = tx * TILE + (wave & 1) * 8, by0 = ty * TILE + (wave >> 1) * 8;
    const int px = bx0 + (lane & 7), py = by0 + (lane >> 3);
    const bool inside = px < d.W && py < d.H;
    const float pxf = (float)px, pyf = (float)py;
    const float bxlo = (float)bx0, bxhi = (float)(bx0 + 7), bylo = (float)by0, byhi = (float)(by0 + 7);

    const int n = (int)s.tile_count[(size_t)g * d.T + tile];
    const uint32_t start = s.tile_end[(size_t)g * d.T + tile] - (uint32_t)n;
    const uint32_t* __restrict__ plist = s.point_list + (size_t)g * d.Rcap + start;
    const uint32_t* __restrict__ slist = s.slot_list + (size_t)g * d.Rcap + start;
    const GeomRec* __restrict__ geom = s.geom + (size_t)g * d.Pa;
    float* const Gout = s.G + (size_t)vin[0] * d.Rcap * G_STRIDE;
    uint8_t* const Gmark = s.row_epoch + (size_t)vin[0] * d.Rcap;
    const unsigned long long* const hm = s.hit_masks ? s.hit_masks + ((size_t)g * hit_mask_words(d.Rcap, d.T) + (start >> 6) + (uint32_t)tile) * 4 : nullptr;

    // per-pixel state
    float T_final = 0.0f;
    uint32_t last_contributor = 0;
    float dpx0 = 0.0f, dpx1 = 0.0f, dpx2 = 0.0f, tfbg = 0.0f, res2[F];
#pragma unroll
    for (int q = 0; q < F; q++) res2[q] = 0.0f;
    if (inside) {
        const size_t pix = (size_t)py * d.W + px;
        T_final = s.final_T[(size_t)g * d.N + pix];
        last_contributor = s.n_contrib[(size_t)g * d.N + pix];
#pragma unroll
        for (int q = 0; q < F; q++) {
            const int v = vin[q];
            float r0, r1, r2;
            if (s.dL_dpix) {
                const float* gp = s.dL_dpix + (size_t)v * 3 * d.N;
                r0 = gp[pix]; r1 = gp[(size_t)d.N + pix]; r2 = gp[2 * (size_t)d.N + pix];
            } else {
                // imageIntToLoss, src/Trainer.cu:33-44: truth/255 - rasterized
                const uint32_t t = s.truth[(size_t)v * d.N + pix];
                const float* out = s.out_color + (size_t)v * 3 * d.N;
                r0 = ((float)(t & 0xFF) / 255.0f) - out[pix];
                r1 = ((float)((t >> 8) & 0xFF) / 255.0f) - out[(size_t)d.N + pix];
                r2 = ((float)((t >> 16) & 0xFF) / 255.0f) - out[2 * (size_t)d.N + pix];
                res2[q] = r0 * r0 + r1 * r1 + r2 * r2;
            }
            const float* bg = s.views[v].bg;
            dpx0 += r0; dpx1 += r1; dpx2 += r2;
            tfbg += -T_final * (bg[0] * r0 + bg[1] * r1 + bg[2] * r2);
        }
    }
    if (tid == 0) sMaxLast = 0;
    __syncthreads();
    // wave-uniform and block-uniform bounds on the traversal
    uint32_t wave_max_last = last_contributor;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wave_max_last = max(wave_max_last, (uint32_t)__shfl_xor((int)wave_max_last, o));
    wave_max_last = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_max_last);  // uniform: branches on it become scalar
    if (lane == 0) atomicMax(&sMaxLast, wave_max_last);
    if (s.loss && !s.dL_dpix) {
#pragma unroll
        for (int q = 0; q < F; q++) {
            const float l = wave_sum_to_lane63(res2[q]);
            if (lane == 63) sLoss[q][wave] = l;
        }
    }
    // dL/dpixel of the pixels this lane contracts (PARK_PPL consecutive pixels of one row), through the parking area
    float* const park = &sPark[wave * PARK_BATCH * PARK_STRIDE];
    float dpr[3][PARK_PPL];
    const int psub = lane % PARK_LPH;
    park[lane] = dpx0; park[64 + lane] = dpx1; park[128 + lane] = dpx2;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int i = 0; i < PARK_PPL; i++) dpr[c][i] = park[c * 64 + psub * PARK_PPL + i];
    const float pyrow = (float)(by0 + ((psub * PARK_PPL) >> 3));
    const float pxcol0 = (float)(bx0 + ((psub * PARK_PPL) & 7));
    __syncthreads();
    const int max_last = (int)sMaxLast;
    if (s.loss && !s.dL_dpix && tid < F)  // fixed summation order: the loss statistic is reproducible too
        s.loss[(size_t)vin[tid] * d.T + tile] = (sLoss[tid][0] + sLoss[tid][1]) + (sLoss[tid][2] + sLoss[tid][3]);
    if (n == 0) return;
    const int rounds = (max_last + ROUND - 1) / ROUND;
    // Entries no pixel reaches have an all-zero gradient row.  With row marks (d.epoch != 0) such rows are neither written here
    // nor read by the per-splat kernel: a row exists iff its mark (row_epoch[slot]) equals this launch's epoch, and only the
    // flush below sets marks.  (Through round 3 every entry got its row: in a dense scene — 1100 entries per tile of which the
    // pixels saturate after ~170 — 84 % of the rows were zeros, 2.2 GB written here and read again by k_splat_bwd_view.)
    // Without marks (cameras whose longest tile list is short: nearly every row exists and the marks would only cost; the
    // decision is taken per camera from the tile scan of this very step, gs_internal.h uses_row_marks) every entry owns a row.
    const bool marks = uses_row_marks(d, s, g);
    if (!marks)
        for (int e = rounds * ROUND + tid; e < n; e += WG) {
            Row3* row = reinterpret_cast<Row3*>(Gout + (size_t)slist[e] * G_STRIDE);
            row[0] = Row3{ 0, 0, 0 }; row[1] = Row3{ 0, 0, 0 }; row[2] = Row3{ 0, 0, 0 };
        }

#ifdef GS_DIAG_COUNT_ACTIVE
    unsigned long long diag_hits = 0, diag_active = 0, diag_staged = 0, diag_pack2 = 0, diag_pack4 = 0;
    unsigned diag_half[2] = { 0, 0 }, diag_quad[4] = { 0, 0, 0, 0 };
#endif
    float T = T_final;
    float arp = 0.0f;  // accum_rec . dL_dpixel, already blended with the previously visited entry
    const float ddelx_dx = 0.5f * (float)d.W, ddely_dy = 0.5f * (float)d.H;
    float* const acc = &sAcc[wave * ROUND * ACC_STRIDE];

    // Staging is spread over the workgroup and runs one round ahead: wave w < 3 fetches 16-byte part w of the record of slot
    // `lane` (slot s <- entry ROUND-1-s of the round), wave 3 the entry's gradient-row slot.  The record of round r-1 is
    // requested right after round r has been staged and arrives during round r's hit loop; its index was requested a round
    // earlier still.  Only the first round of a tile pays the two dependent global latencies.
    // (Every round below the first one walked is full; the first may be the list's partial tail.)
    auto entry_of = [&](int r) { return r * ROUND + ROUND - 1 - lane; };  // this lane's entry in round r
    const bool first_valid = rounds > 0 && entry_of(rounds - 1) < n;
    uint32_t idx_next = 0;   // waves 0-2: splat id of this lane's entry in the NEXT round to stage; wave 3: its row slot
    float4 part = make_float4(0, 0, 0, 0);
    {
        const uint32_t* __restrict__ src = wave < 3 ? plist : slist;
        const uint32_t idx0 = first_valid ? src[entry_of(rounds - 1)] : 0u;
        if (rounds > 1) idx_next = src[entry_of(rounds - 2)];
        if (wave < 3) { if (first_valid) part = reinterpret_cast<const float4*>(geom + idx0)[wave]; }
        else part.x = __uint_as_float(idx0);
    }

    // the forward's ballot of the first round walked (only rounds this wave still has entries to visit in are ever used)
    unsigned long long ballot_next = (hm && rounds > 0) ? hm[(size_t)(rounds - 1) * 4 + wave] : 0ull;
    for (int r = rounds - 1; r >= 0; r--) {
        const int base = r * ROUND;
        const int cnt = min(ROUND, n - base);
        __syncthreads();  // previous round's flush has consumed st / sAcc / sTouched
        if (wave == 0) { part.z *= -0.5f * LOG2E; part.w *= -LOG2E; st.A[lane] = part; }   // conic pre-scaled to base 2 (stage_entry)
        else if (wave == 1) { part.x *= -0.5f * LOG2E; st.B[lane] = part; }
        else if (wave == 2) { part.y *= LOG2E; st.C[lane] = part; }
        else sSlot[lane] = __float_as_uint(part.x);
        __syncthreads();
        if (r > 0) {  // request round r-1 (always a full round)
            if (wave < 3) part = reinterpret_cast<const float4*>(geom + idx_next)[wave];
            else part.x = __uint_as_float(idx_next);
            if (r > 1) idx_next = (wave < 3 ? plist : slist)[entry_of(r - 2)];
        }
        unsigned long long mask;
        if (hm) {
            // the forward of this camera tested the same 64 entries against the same block: its ballot (bit j = entry j of the
            // sub-block), reversed into slot order and cut to the entries this wave still has to visit.  (The forward computed
            // it for every sub-block in which a pixel of the wave was still alive, which includes the one holding the wave's last
            // contributor and all before it.)
            const int nvalid = max(0, min(cnt, (int)min((uint32_t)ROUND, wave_max_last - min(wave_max_last, (uint32_t)base))));
            mask = nvalid > 0 ? (__builtin_bitreverse64(ballot_next) & (~0ull << (ROUND - nvalid))) : 0ull;
            if (r > 0) ballot_next = hm[(size_t)(r - 1) * 4 + wave];  // requested a round ahead, like the staging
        } else {
            bool hit = false;
            const int j = ROUND - 1 - lane;
            if (j < cnt && (uint32_t)(base + j) < wave_max_last) {
                const float4 a = st.A[lane];
                hit = block_reaches(a.x, a.y, -2.0f * a.z, -a.w, -2.0f * st.B[lane].x, st.C[lane].y, bxlo, bxhi, bylo, byhi);
            }
            mask = __ballot(hit);
        }
#ifdef GS_DIAG_NO_HITS  // timing experiments only (tools/build_variant.sh): the round skeleton without the hit loop
        mask = 0ull;
#endif
        const unsigned long long touched = mask;
#ifdef GS_DIAG_COUNT_ACTIVE
        diag_staged += (unsigned long long)max(0, min(cnt, (int)min((uint32_t)ROUND, wave_max_last - min(wave_max_last, (uint32_t)base))));
#endif
        const uint32_t pos_slot0 = (uint32_t)(base + ROUND - 1);  // upstream's `contributor` (after its decrement) of slot 0
        // Hits are taken PARK_BATCH at a time: the inner loop is unrolled, so a hit's parking offset is an immediate and the
        // batch needs no counter arithmetic; the round's last, partly filled batch leaves the inner loop early.
        while (mask != 0ull) {
            int nb = 0;          // hits parked in this batch
            uint32_t blo = 0;    // their slots, 8 bits apiece, newest in the low byte
#pragma unroll
            for (int hq = 0; hq < PARK_BATCH; hq++) {
                if (mask == 0ull) break;
                const int k = __builtin_ctzll(mask);
                asm("s_bitset0_b64 %0, %1" : "+s"(mask) : "s"(k));
                const float4 Ac = st.A[k], Bc = st.B[k];
                const float cbc = st.C[k].x;
                const uint32_t pos = pos_slot0 - (uint32_t)k;
                const float dx = Ac.x - pxf, dy = Ac.y - pyf;
                const float power = dx * (Ac.z * dx + Ac.w * dy) + Bc.x * dy * dy;  // log2 of the Gaussian weight
                const float G0 = __builtin_amdgcn_exp2f(power);
                const float alpha0 = fminf(ALPHA_MAX, Bc.y * G0);
                const bool act = (pos < last_contributor) & (power <= 0.0f) & (alpha0 >= ALPHA_MIN);
#ifdef GS_DIAG_COUNT_ACTIVE
                {
                    const unsigned long long am = __ballot(act);
                    diag_hits++; diag_active += (unsigned long long)__popcll(am);
                    // what packing entries at a finer granularity could reclaim: hits that reach the upper / lower 8x4 half and each of
                    // the four 4x4 quadrants of the block (lane = 8 y + x)
                    diag_half[0] += (am & 0x00000000FFFFFFFFull) != 0; diag_half[1] += (am & 0xFFFFFFFF00000000ull) != 0;
                    diag_quad[0] += (am & 0x000000000F0F0F0Full) != 0; diag_quad[1] += (am & 0x00000000F0F0F0F0ull) != 0;
                    diag_quad[2] += (am & 0x0F0F0F0F00000000ull) != 0; diag_quad[3] += (am & 0xF0F0F0F000000000ull) != 0;
                }
#endif
                const float G = act ? G0 : 0.0f, alpha = act ? alpha0 : 0.0f;
                const float inv1ma = __builtin_amdgcn_rcpf(1.0f - alpha);  // exactly 1 for an inactive lane
                T = T * inv1ma;
                const float w = alpha * T;
                // upstream: dL_dalpha = sum_ch (colour_ch - accum_rec_ch) * dL_dpixel_ch with accum_rec blended per channel,
                // accum_rec = last_alpha * last_colour + (1 - last_alpha) * accum_rec.  Only the contraction with dL_dpixel is ever
                // used, and it obeys the same recurrence: arp = sum_ch accum_rec_ch * dL_dpixel_ch  ->  arp += alpha * (cd - arp)
                // with cd = sum_ch colour_ch * dL_dpixel_ch (blend applied AFTER use: the identical recurrence one step early).
                const float cd = Bc.z * dpx0 + Bc.w * dpx1 + cbc * dpx2;
                const float dL = cd - arp;
                arp = fmaf(alpha, dL, arp);
                const float u = G * (dL * T + tfbg * inv1ma);  // G * (dL_dalpha * T + tfbg / (1 - alpha)),  tfbg = -T_final * (bg . dL_dpix)
                park[hq * PARK_STRIDE + lane] = w;
                park[hq * PARK_STRIDE + 64 + lane] = u;
                blo = (blo << 8) | (uint32_t)k;
                nb = hq + 1;
            }
            __builtin_amdgcn_wave_barrier();
#ifndef GS_DIAG_NO_CONTRACT
            contract_parked(park, st, acc, nb, blo, dpr, pxcol0, pyrow, lane);
#endif
            __builtin_amdgcn_wave_barrier();
        }
        if (lane == 0) sTouched[wave] = touched;
#ifdef GS_DIAG_COUNT_ACTIVE
        // iterations of a round if its hits were packed two (8x4 halves) / four (4x4 quadrants) entries to a wave iteration
        diag_pack2 += max(diag_half[0], diag_half[1]);
        diag_pack4 += max(max(diag_quad[0], diag_quad[1]), max(diag_quad[2], diag_quad[3]));
        diag_half[0] = diag_half[1] = 0; diag_quad[0] = diag_quad[1] = diag_quad[2] = diag_quad[3] = 0;
        if (r == 0 && lane == 0) {
            atomicAdd(&g_diag_counters[2], diag_hits); atomicAdd(&g_diag_counters[3], diag_active); atomicAdd(&g_diag_counters[5], diag_staged);
            atomicAdd(&g_diag_counters[6], diag_pack2); atomicAdd(&g_diag_counters[7], diag_pack4);
        }
#endif
        __syncthreads();
        // a slot no wave evaluated this round (no block of the tile can reach alpha >= 1/255 there) keeps its implicit zero row
        const bool any_hit = !marks || ((((sTouched[0] | sTouched[1]) | (sTouched[2] | sTouched[3])) >> lane) & 1ull);
        if (wave < 3 && ROUND - 1 - lane < cnt && any_hit) {
            // moments -> the reference's nine sums (dx = mean2D.x - pixel.x as upstream):
            //   dL_dmean2D.x = -0.5 W op (conA * S[u dx] + conB * S[u dy]),  .y = -0.5 H op (conC * S[u dy] + conB * S[u dx])
            //   dL_dconic    = -0.5 op (S[u dx dx], S[u dx dy], S[u dy dy]),  dL_dopacity = S[u]
            // Wave w < 3 finishes 12-byte group w of every slot's row (colour | mean2D + conic.x | conic.y, conic.z, opacity):
            // a third of the LDS reads and one store per thread instead of the whole row on wave 0 while three waves wait.
            float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
#pragma unroll
            for (int w = 0; w < 4; w++) {  // fixed order over the four waves; only slots written this round are read
                if ((sTouched[w] >> lane) & 1ull) {
                    const float* a = &sAcc[(w * ROUND + lane) * ACC_STRIDE + 3 * wave];
                    s0 += a[0]; s1 += a[1]; s2 += a[2];
                }
            }
            Row3 out{ s0, s1, s2 };
            if (wave != 0) {
                const float4 Af = st.A[lane], Bf = st.B[lane];
                const float op = Bf.y, hop = -0.5f * op;
                if (wave == 1) {
                    float conA, conB, conC;
                    unscaled_conic(Af, Bf, conA, conB, conC);
                    out = Row3{ -ddelx_dx * op * (conA * s0 + conB * s1), -ddely_dy * op * (conC * s1 + conB * s0), hop * s2 };
                } else {
                    out = Row3{ hop * s0, hop * s1, s2 };
                }
            }
#ifdef GS_DIAG_ROWS_IN_TILE_ORDER  // timing experiment: rows at the entry's list position (coalesced 2304-byte runs) instead of its slot
            Row3* row = reinterpret_cast<Row3*>(Gout + ((size_t)start + base + (ROUND - 1 - lane)) * G_STRIDE);
#else
            Row3* row = reinterpret_cast<Row3*>(Gout + (size_t)sSlot[lane] * G_STRIDE);
#endif
#ifndef GS_DIAG_NO_ROWS  // timing experiment: the kernel without its gradient-row stores
            row[wave] = out;
            if (wave == 0 && marks) Gmark[sSlot[lane]] = (uint8_t)d.epoch;
#endif
        }
    }
}

__global__ __launch_bounds__(WG) void k_loss_sum(Dims d, Scratch s) {
    __shared__ float part[WG];
    const int v = blockIdx.x;
    const float* src = s.loss + (size_t)v * d.T;
    float acc = 0.0f;
    for (int t = threadIdx.x; t < d.T; t += WG) acc += src[t];
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int o = WG / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) s.loss_total[v] = part[0];
}
int launch_loss_sum(const Dims& d, const Scratch& s, hipStream_t stream) {
    if (d.V == 0 || !s.loss || !s.loss_total) return GS_OK;
    hipLaunchKernelGGL(k_loss_sum, dim3(d.V), dim3(WG), 0, stream, d, s);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

// blockIdx.y enumerates passes for the per-pass form — the two passes of every pair item first, then the single items
// (the order k_splat_bwd_view reads the rows in) — and pair items for the fused form.
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(GS_BWD_PAIR_WAVES, GS_BWD_PAIR_WAVES))) void k_render_bwd1(Dims d, Scratch s, const int* __restrict__ items, int n_pairs) {
    const int y = blockIdx.y;
    const int* it = items + 3 * (y < 2 * n_pairs ? (y >> 1) : (y - n_pairs));
    const int one[2] = { it[0], it[1 + (y < 2 * n_pairs ? (y & 1) : 0)] };
    render_bwd<1>(d, s, one);
}
__global__ __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(GS_BWD_PAIR_WAVES, GS_BWD_PAIR_WAVES))) void k_render_bwd_pair(Dims d, Scratch s, const int* __restrict__ items) {
    render_bwd<2>(d, s, items + 3 * blockIdx.y);
}

// items: n2 pairs {group, pass a, pass b} followed by n1 singles {group, pass, -1} (device array of 3*(n2+n1) ints).
// fuse_pairs: one gradient set per pair (rows in pass a's slice of G) instead of one per pass.
int launch_render_backward(const Dims& d, const Scratch& s, const int* items, int n2, int n1, bool fuse_pairs, hipStream_t stream) {
    if (d.T == 0) return GS_OK;
    if (n2 > 0 && fuse_pairs) {
        hipLaunchKernelGGL(k_render_bwd_pair, dim3(d.T, n2), dim3(WG), 0, stream, d, s, items);
        if (n1 > 0) hipLaunchKernelGGL(k_render_bwd1, dim3(d.T, n1), dim3(WG), 0, stream, d, s, items + 3 * n2, 0);
    } else if (2 * n2 + n1 > 0) {
        hipLaunchKernelGGL(k_render_bwd1, dim3(d.T, 2 * n2 + n1), dim3(WG), 0, stream, d, s, items, n2);
    }
    GS_HIP(hipGetLastError());
    return GS_OK;
}

}  // namespace gs
