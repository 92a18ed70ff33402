// k_render.hip — forward alpha-compositing raster and per-pixel backward pass.
// Replaces renderCUDA (forward) and renderCUDA (backward) inside CudaRasterizer::Rasterizer::forward /
// ::backward (reference call sites src/Trainer.cu:334-360, :378-412)
// and fuses the reference's imageIntToLoss (src/Trainer.cu:33-44) into the backward prologue.
//
// MI355X mapping (not the upstream one):
//   * workgroup = one 16x16 tile = 4 wave64; each WAVE owns an 8x8 pixel block, lane = pixel.
//   * the tile's depth-ordered list is staged through LDS 256 entries at a time (48-byte records
//     gathered from the 64-byte per-splat lines written by preprocess).
//   * per 64 staged entries every lane tests ONE entry's "alpha >= 1/255" box against the wave's
//     8x8 block; the ballot is a scalar bit list and only surviving entries are evaluated — the
//     skipped pairs are exactly ones the reference blend would skip too, so results are unchanged.
//   * backward: the nine per-pair partial derivatives are reduced across the 64 lanes with DPP row
//     operations (no LDS, no atomics), each wave parks its sum in its own LDS slot, the four slots
//     are added in fixed order and written as ONE 48-byte row per (splat,tile) entry.  There is no
//     global atomic anywhere in the backward pass and the result is bitwise reproducible.
#include "gs_internal.h"

namespace gs {

constexpr float ALPHA_MIN = 1.0f / 255.0f;
constexpr float ALPHA_MAX = 0.99f;
constexpr float T_STOP = 0.0001f;
constexpr int ACC_STRIDE = 9;

__device__ inline bool overlaps(float x, float y, float hx, float hy, float bxlo, float bxhi, float bylo, float byhi) {
    return (x + hx >= bxlo) && (x - hx <= bxhi) && (y + hy >= bylo) && (y - hy <= byhi);
}

struct StagedTile {
    float4 A[WG];   // x, y, conA, conB
    float4 B[WG];   // conC, opacity, r, g
    float C[WG];    // b
    float2 Hh[WG];  // hx, hy
};

__device__ inline void stage_entry(StagedTile& t, int slot, const GeomRec* __restrict__ r) {
    const float4* q = reinterpret_cast<const float4*>(r);
    const float4 a = q[0], b = q[1], c = q[2];
    t.A[slot] = a; t.B[slot] = b; t.C[slot] = c.x; t.Hh[slot] = make_float2(c.y, c.z);
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WG) void k_render_fwd(Dims d, Scratch s) {
    __shared__ StagedTile st;
    const int tile = blockIdx.x, v = blockIdx.y;
    if (s.flags[v * 4 + 0] & 1u) return;
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

    float T = 1.0f, C0 = 0.0f, C1 = 0.0f, C2 = 0.0f;
    uint32_t last = 0;
    bool done = !inside;

    for (int base = 0; base < n; base += WG) {
        if (__syncthreads_and(done)) break;  // whole tile saturated; also guards LDS reuse
        const int e = base + tid;
        if (e < n) stage_entry(st, tid, geom + plist[e]);
        __syncthreads();
        const int cnt = min(WG, n - base);
        for (int sub = 0; sub < cnt; sub += 64) {
            if (__ballot(!done) == 0ull) break;
            const int j = sub + lane;
            bool hit = false;
            if (j < cnt) {
                const float4 a = st.A[j];
                const float2 h = st.Hh[j];
                hit = overlaps(a.x, a.y, h.x, h.y, bxlo, bxhi, bylo, byhi);
            }
            unsigned long long mask = __ballot(hit);
            while (mask) {
                const int k = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const int jj = sub + k;
                const float4 A = st.A[jj];
                const float4 B = st.B[jj];
                const float cb = st.C[jj];
                const float dx = A.x - pxf, dy = A.y - pyf;
                const float power = -0.5f * (A.z * dx * dx + B.x * dy * dy) - A.w * dx * dy;
                const float alpha = fminf(ALPHA_MAX, B.y * __expf(power));
                if (!done && power <= 0.0f && alpha >= ALPHA_MIN) {
                    const float test_T = T * (1.0f - alpha);
                    if (test_T < T_STOP) done = true;
                    else {
                        C0 += B.z * alpha * T; C1 += B.w * alpha * T; C2 += cb * alpha * T;
                        T = test_T;
                        last = (uint32_t)(base + jj + 1);
                    }
                }
            }
        }
    }
    if (inside) {
        const size_t pix = (size_t)py * d.W + px;
        const float* bg = s.views[v].bg;
        float* out = s.out_color + (size_t)v * 3 * d.N;
        s.final_T[(size_t)v * d.N + pix] = T;
        s.n_contrib[(size_t)v * d.N + pix] = last;
        out[pix] = C0 + T * bg[0];
        out[(size_t)d.N + pix] = C1 + T * bg[1];
        out[2 * (size_t)d.N + pix] = C2 + T * bg[2];
    }
}

int launch_render_forward(const Dims& d, const Scratch& s, hipStream_t stream) {
    if (d.T == 0 || d.V == 0) return GS_OK;
    hipLaunchKernelGGL(k_render_fwd, dim3(d.T, d.V), dim3(WG), 0, stream, d, s);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

// ---------------------------------------------------------------------------------------------
// backward
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

__global__ __launch_bounds__(WG) void k_render_bwd(Dims d, Scratch s) {
    __shared__ StagedTile st;
    __shared__ uint32_t sSlot[WG];
    __shared__ float sAcc[4 * WG * ACC_STRIDE];
    __shared__ uint32_t sMaxLast;
    __shared__ float sLoss;
    const int tile = blockIdx.x, v = blockIdx.y;
    if (s.flags[v * 4 + 0] & 1u) return;
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
    const uint32_t* __restrict__ slist = s.slot_list + (size_t)v * d.Rcap + start;
    const GeomRec* __restrict__ geom = s.geom + (size_t)v * d.Pa;
    float* __restrict__ Gv = s.G + (size_t)v * d.Rcap * G_STRIDE;
    const float* bg = s.views[v].bg;

    // per-pixel state
    float T_final = 0.0f;
    uint32_t last_contributor = 0;
    float dpx0 = 0.0f, dpx1 = 0.0f, dpx2 = 0.0f;
    float res2 = 0.0f;
    if (inside) {
        const size_t pix = (size_t)py * d.W + px;
        T_final = s.final_T[(size_t)v * d.N + pix];
        last_contributor = s.n_contrib[(size_t)v * d.N + pix];
        if (s.dL_dpix) {
            const float* g = s.dL_dpix + (size_t)v * 3 * d.N;
            dpx0 = g[pix]; dpx1 = g[(size_t)d.N + pix]; dpx2 = g[2 * (size_t)d.N + pix];
        } else {
            // imageIntToLoss, src/Trainer.cu:33-44: truth/255 - rasterized
            const uint32_t t = s.truth[(size_t)v * d.N + pix];
            const float* out = s.out_color + (size_t)v * 3 * d.N;
            dpx0 = ((float)(t & 0xFF) / 255.0f) - out[pix];
            dpx1 = ((float)((t >> 8) & 0xFF) / 255.0f) - out[(size_t)d.N + pix];
            dpx2 = ((float)((t >> 16) & 0xFF) / 255.0f) - out[2 * (size_t)d.N + pix];
            res2 = dpx0 * dpx0 + dpx1 * dpx1 + dpx2 * dpx2;
        }
    }
    if (tid == 0) { sMaxLast = 0; sLoss = 0.0f; }
    for (int k = tid; k < 4 * WG * ACC_STRIDE; k += WG) sAcc[k] = 0.0f;
    __syncthreads();
    // wave-uniform and block-uniform bounds on the traversal
    uint32_t wave_max_last = last_contributor;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wave_max_last = max(wave_max_last, (uint32_t)__shfl_xor((int)wave_max_last, o));
    if (lane == 0) atomicMax(&sMaxLast, wave_max_last);
    if (s.loss && !s.dL_dpix) {
        const float l = wave_sum_to_lane63(res2);
        if (lane == 63) atomicAdd(&sLoss, l);
    }
    __syncthreads();
    const int max_last = (int)sMaxLast;
    if (s.loss && !s.dL_dpix && tid == 0) atomicAdd(&s.loss[v], sLoss);
    if (n == 0) return;
    const int rounds = (max_last + WG - 1) / WG;
    // entries no pixel reaches still own a gradient row: zero it
    for (int p = rounds * WG + tid; p < n; p += WG) {
        float4* row = reinterpret_cast<float4*>(Gv + (size_t)slist[p] * G_STRIDE);
        row[0] = make_float4(0, 0, 0, 0); row[1] = make_float4(0, 0, 0, 0); row[2] = make_float4(0, 0, 0, 0);
    }

    float T = T_final;
    float ar0 = 0.0f, ar1 = 0.0f, ar2 = 0.0f;  // accum_rec
    float lc0 = 0.0f, lc1 = 0.0f, lc2 = 0.0f;  // last_color
    float last_alpha = 0.0f;
    const float bg_dot = bg[0] * dpx0 + bg[1] * dpx1 + bg[2] * dpx2;
    const float ddelx_dx = 0.5f * (float)d.W, ddely_dy = 0.5f * (float)d.H;

    for (int r = rounds - 1; r >= 0; r--) {
        const int base = r * WG;
        const int cnt = min(WG, n - base);
        __syncthreads();  // previous round's flush has consumed st / sAcc
        if (tid < cnt) {
            stage_entry(st, tid, geom + plist[base + tid]);
            sSlot[tid] = slist[base + tid];
        }
        __syncthreads();
        for (int sub = ((cnt - 1) >> 6) << 6; sub >= 0; sub -= 64) {
            if ((uint32_t)(base + sub) >= wave_max_last) continue;
            const int j = sub + lane;
            bool hit = false;
            if (j < cnt && (uint32_t)(base + j) < wave_max_last) {
                const float4 a = st.A[j];
                const float2 h = st.Hh[j];
                hit = overlaps(a.x, a.y, h.x, h.y, bxlo, bxhi, bylo, byhi);
            }
            unsigned long long mask = __ballot(hit);
            while (mask) {
                const int k = 63 - __clzll((long long)mask);
                mask &= ~(1ull << k);
                const int jj = sub + k;
                const uint32_t pos = (uint32_t)(base + jj);  // upstream's `contributor` after its decrement
                const float4 A = st.A[jj];
                const float4 B = st.B[jj];
                const float cb = st.C[jj];
                const float dx = A.x - pxf, dy = A.y - pyf;
                const float power = -0.5f * (A.z * dx * dx + B.x * dy * dy) - A.w * dx * dy;
                const float G = __expf(power);
                const float alpha = fminf(ALPHA_MAX, B.y * G);
                const bool act = (pos < last_contributor) && power <= 0.0f && alpha >= ALPHA_MIN;
                float g0 = 0, g1 = 0, g2 = 0, g3 = 0, g4 = 0, g5 = 0, g6 = 0, g7 = 0, g8 = 0;
                if (act) {
                    const float inv1ma = __builtin_amdgcn_rcpf(1.0f - alpha);
                    T = T * inv1ma;
                    const float dchannel_dcolor = alpha * T;
                    float dL_dalpha = 0.0f;
                    ar0 = last_alpha * lc0 + (1.0f - last_alpha) * ar0; lc0 = B.z; dL_dalpha += (B.z - ar0) * dpx0; g0 = dchannel_dcolor * dpx0;
                    ar1 = last_alpha * lc1 + (1.0f - last_alpha) * ar1; lc1 = B.w; dL_dalpha += (B.w - ar1) * dpx1; g1 = dchannel_dcolor * dpx1;
                    ar2 = last_alpha * lc2 + (1.0f - last_alpha) * ar2; lc2 = cb; dL_dalpha += (cb - ar2) * dpx2; g2 = dchannel_dcolor * dpx2;
                    dL_dalpha *= T;
                    last_alpha = alpha;
                    dL_dalpha += (-T_final * inv1ma) * bg_dot;
                    const float dL_dG = B.y * dL_dalpha;
                    const float gdx = G * dx, gdy = G * dy;
                    const float dG_ddelx = -gdx * A.z - gdy * A.w;
                    const float dG_ddely = -gdy * B.x - gdx * A.w;
                    g3 = dL_dG * dG_ddelx * ddelx_dx;
                    g4 = dL_dG * dG_ddely * ddely_dy;
                    g5 = -0.5f * gdx * dx * dL_dG;
                    g6 = -0.5f * gdx * dy * dL_dG;
                    g7 = -0.5f * gdy * dy * dL_dG;
                    g8 = G * dL_dalpha;
                }
                if (__ballot(act) != 0ull) {
                    g0 = wave_sum_to_lane63(g0); g1 = wave_sum_to_lane63(g1); g2 = wave_sum_to_lane63(g2);
                    g3 = wave_sum_to_lane63(g3); g4 = wave_sum_to_lane63(g4); g5 = wave_sum_to_lane63(g5);
                    g6 = wave_sum_to_lane63(g6); g7 = wave_sum_to_lane63(g7); g8 = wave_sum_to_lane63(g8);
                    if (lane == 63) {
                        float* a = &sAcc[(wave * WG + jj) * ACC_STRIDE];
                        a[0] = g0; a[1] = g1; a[2] = g2; a[3] = g3; a[4] = g4; a[5] = g5; a[6] = g6; a[7] = g7; a[8] = g8;
                    }
                }
            }
        }
        __syncthreads();
        if (tid < cnt) {
            float sum[ACC_STRIDE];
#pragma unroll
            for (int q = 0; q < ACC_STRIDE; q++) {
                float acc = 0.0f;
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    float* a = &sAcc[(w * WG + tid) * ACC_STRIDE + q];
                    acc += *a;
                    *a = 0.0f;
                }
                sum[q] = acc;
            }
            float4* row = reinterpret_cast<float4*>(Gv + (size_t)sSlot[tid] * G_STRIDE);
            row[0] = make_float4(sum[0], sum[1], sum[2], sum[3]);
            row[1] = make_float4(sum[4], sum[5], sum[6], sum[7]);
            row[2] = make_float4(sum[8], 0.0f, 0.0f, 0.0f);
        }
    }
}

int launch_render_backward(const Dims& d, const Scratch& s, hipStream_t stream) {
    if (d.T == 0 || d.V == 0) return GS_OK;
    hipLaunchKernelGGL(k_render_bwd, dim3(d.T, d.V), dim3(WG), 0, stream, d, s);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

}  // namespace gs
