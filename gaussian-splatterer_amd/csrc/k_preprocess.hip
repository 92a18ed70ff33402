// k_preprocess.hip — per-(view,splat) projection: frustum cull, cov3D, EWA cov2D, conic, radius,
// tile rect, SH colour (+ its direction Jacobian for the trainer's backward); per-super-tile counting in an LDS
// histogram, the block's row of the binning's count matrix and the block's tile sum.  Replaces the preprocess stage of
// CudaRasterizer::Rasterizer::forward (reference call site src/Trainer.cu:334-360; algorithm
// SURVEY.md Appendix A.1).  One thread per splat, blockIdx.y = view; SoA parameter planes give
// 256-byte coalesced wave loads.  Built with -ffp-contract=off: every fp32 operation here is an
// individually rounded IEEE operation in the same order as oracle/gs_oracle.cpp::preprocess, so
// depth, radius and tile rectangle (and therefore the sorted tile lists) are bit-identical.
#include <hip/hip_fp16.h>

#include <algorithm>

#include "gs_internal.h"
#include "sh_jac.h"

namespace gs {

__device__ __constant__ float SH_C0 = 0.28209479177387814f;
__device__ __constant__ float SH_C1 = 0.4886025119029199f;
__device__ __constant__ float SH_C2[5] = { 1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                           -1.0925484305920792f, 0.5462742152960396f };
__device__ __constant__ float SH_C3[7] = { -0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                                           0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                                           -0.5900435899266435f };

// The parameters of one splat, loaded once and projected through every camera of the block's chunk: a thread per (camera,
// splat) re-read the 11 + 3M planes per camera and spent its life waiting for them (0.060 ms per launch at cfg3 for 9 us of
// arithmetic).
template <int D> struct SplatIn {
    static constexpr int NC = (D + 1) * (D + 1);
    float loc[3], scale[3], rot[4], opacity;
    float sh[NC][3];
};
template <int D, bool H>
__device__ inline void load_splat(const Dims& d, const float* __restrict__ params, const Scratch& s, int i, SplatIn<D>& in) {
    const Planes pl{ d.M };
    const size_t st = (size_t)d.Pa;
#pragma unroll
    for (int c = 0; c < 3; c++) { in.loc[c] = params[pl.loc(c) * st + i]; in.scale[c] = params[pl.scale(c) * st + i]; }
#pragma unroll
    for (int c = 0; c < 4; c++) in.rot[c] = params[pl.rot(c) * st + i];
    in.opacity = params[pl.opac() * st + i];
    const __half* sh16 = reinterpret_cast<const __half*>(s.sh16);
#pragma unroll
    for (int k = 0; k < SplatIn<D>::NC; k++)
#pragma unroll
        for (int c = 0; c < 3; c++) {
            if constexpr (H) in.sh[k][c] = __half2float(sh16[(size_t)(3 * k + c) * st + i]);
            else in.sh[k][c] = params[pl.sh(k, c) * st + i];
        }
}

// one (view, splat): writes the record and tiles_touched, counts the splat into its super-tiles; returns tiles_touched
// H: the SH coefficients come from the trainer's half-precision read copy (s.sh16) instead of the fp32 planes; everything
// else — geometry, lists, ranges — does not touch SH and is bit-identical in both modes.
template <int D>
__device__ inline uint32_t preprocess_one(const Dims& d, const SplatIn<D>& in, const Scratch& s, int i, int v, uint32_t* hist) {
    const gs_view& vp = s.gviews[v];  // v indexes geometry groups here
    const size_t st = (size_t)d.Pa;
    GeomRec* rec = s.geom + (size_t)v * st + i;
    uint32_t* tt = s.tiles_touched + (size_t)v * st + i;

    const float px_ = in.loc[0], py_ = in.loc[1], pz_ = in.loc[2];
    const float* vm = vp.view;
    const float* pm = vp.projview;
    // p_view (transformPoint4x3), near-plane cull
    const float pvx = vm[0] * px_ + vm[4] * py_ + vm[8] * pz_ + vm[12];
    const float pvy = vm[1] * px_ + vm[5] * py_ + vm[9] * pz_ + vm[13];
    const float pvz = vm[2] * px_ + vm[6] * py_ + vm[10] * pz_ + vm[14];
    bool visible = !(pvz <= 0.2f);

    float conx = 0, cony = 0, conz = 0, pix_x = 0, pix_y = 0, my_radius = 0;
    int rminx = 0, rminy = 0, rmaxx = 0, rmaxy = 0;
    if (visible) {
        const float phx = pm[0] * px_ + pm[4] * py_ + pm[8] * pz_ + pm[12];
        const float phy = pm[1] * px_ + pm[5] * py_ + pm[9] * pz_ + pm[13];
        const float phw = pm[3] * px_ + pm[7] * py_ + pm[11] * pz_ + pm[15];
        const float p_w = 1.0f / (phw + 0.0000001f);
        const float ppx = phx * p_w, ppy = phy * p_w;

        const float sx = d.mod * in.scale[0], sy = d.mod * in.scale[1], sz = d.mod * in.scale[2];
        const float r = in.rot[0], x = in.rot[1], y = in.rot[2], z = in.rot[3];
        float Rg[3][3];
        Rg[0][0] = 1.0f - 2.0f * (y * y + z * z); Rg[0][1] = 2.0f * (x * y - r * z); Rg[0][2] = 2.0f * (x * z + r * y);
        Rg[1][0] = 2.0f * (x * y + r * z); Rg[1][1] = 1.0f - 2.0f * (x * x + z * z); Rg[1][2] = 2.0f * (y * z - r * x);
        Rg[2][0] = 2.0f * (x * z - r * y); Rg[2][1] = 2.0f * (y * z + r * x); Rg[2][2] = 1.0f - 2.0f * (x * x + y * y);
        float Mm[3][3];
#pragma unroll
        for (int c = 0; c < 3; c++) { Mm[c][0] = sx * Rg[c][0]; Mm[c][1] = sy * Rg[c][1]; Mm[c][2] = sz * Rg[c][2]; }
#define GS_SIG(c, rr) (Mm[rr][0] * Mm[c][0] + Mm[rr][1] * Mm[c][1] + Mm[rr][2] * Mm[c][2])
        const float c30 = GS_SIG(0, 0), c31 = GS_SIG(0, 1), c32 = GS_SIG(0, 2), c33 = GS_SIG(1, 1), c34 = GS_SIG(1, 2),
                    c35 = GS_SIG(2, 2);
#undef GS_SIG
        const float focal_x = (float)d.W / (2.0f * vp.tan_fovx);
        const float focal_y = (float)d.H / (2.0f * vp.tan_fovy);
        const float limx = 1.3f * vp.tan_fovx, limy = 1.3f * vp.tan_fovy;
        const float txtz = pvx / pvz, tytz = pvy / pvz;
        const float tx = fminf(limx, fmaxf(-limx, txtz)) * pvz;
        const float ty = fminf(limy, fmaxf(-limy, tytz)) * pvz;
        const float tz = pvz;
        const float J00 = focal_x / tz, J02 = -(focal_x * tx) / (tz * tz);
        const float J11 = focal_y / tz, J12 = -(focal_y * ty) / (tz * tz);
        float T[2][3];
#pragma unroll
        for (int rr = 0; rr < 3; rr++) {
            T[0][rr] = vm[4 * rr] * J00 + vm[4 * rr + 2] * J02;
            T[1][rr] = vm[4 * rr + 1] * J11 + vm[4 * rr + 2] * J12;
        }
        const float V[3][3] = { { c30, c31, c32 }, { c31, c33, c34 }, { c32, c34, c35 } };
        float A[3][2];
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int rr = 0; rr < 2; rr++) A[k][rr] = T[rr][0] * V[0][k] + T[rr][1] * V[1][k] + T[rr][2] * V[2][k];
        float ca = A[0][0] * T[0][0] + A[1][0] * T[0][1] + A[2][0] * T[0][2];
        const float cb = A[0][1] * T[0][0] + A[1][1] * T[0][1] + A[2][1] * T[0][2];
        float cc = A[0][1] * T[1][0] + A[1][1] * T[1][1] + A[2][1] * T[1][2];
        ca += 0.3f; cc += 0.3f;
        const float det = ca * cc - cb * cb;
        if (det == 0.0f) visible = false;
        else {
            const float det_inv = 1.0f / det;
            conx = cc * det_inv; cony = -cb * det_inv; conz = ca * det_inv;
            const float mid = 0.5f * (ca + cc);
            const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
            const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
            my_radius = ceilf(3.0f * sqrtf(fmaxf(lambda1, lambda2)));
            pix_x = ((ppx + 1.0f) * (float)d.W - 1.0f) * 0.5f;
            pix_y = ((ppy + 1.0f) * (float)d.H - 1.0f) * 0.5f;
            const int max_radius = (int)my_radius;
            rminx = min(d.gx, max(0, (int)((pix_x - (float)max_radius) / (float)TILE)));
            rminy = min(d.gy, max(0, (int)((pix_y - (float)max_radius) / (float)TILE)));
            rmaxx = min(d.gx, max(0, (int)((pix_x + (float)max_radius + (float)(TILE - 1)) / (float)TILE)));
            rmaxy = min(d.gy, max(0, (int)((pix_y + (float)max_radius + (float)(TILE - 1)) / (float)TILE)));
            if ((rmaxx - rminx) * (rmaxy - rminy) == 0) visible = false;
        }
    }
    if (!visible) {
        rec->radius = 0;
        *tt = 0;
        return 0u;
    }

    // colour from SH
    float dx = px_ - vp.campos[0], dy = py_ - vp.campos[1], dz = pz_ - vp.campos[2];
    const float len = sqrtf(dx * dx + dy * dy + dz * dz);
    dx = dx / len; dy = dy / len; dz = dz / len;
    float res[3], jac[9];
    uint32_t flags = 0;
    auto shv = [&](int k, int c) -> float { return in.sh[k][c]; };
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float val = SH_C0 * shv(0, c);
        if (D > 0) {
            const float X = dx, Y = dy, Z = dz;
            val = val - SH_C1 * Y * shv(1, c) + SH_C1 * Z * shv(2, c) -
                  SH_C1 * X * shv(3, c);
            if (D > 1) {
                const float xx = X * X, yy = Y * Y, zz = Z * Z, xy = X * Y, yz = Y * Z, xz = X * Z;
                val = val + SH_C2[0] * xy * shv(4, c) + SH_C2[1] * yz * shv(5, c) +
                      SH_C2[2] * (2.0f * zz - xx - yy) * shv(6, c) +
                      SH_C2[3] * xz * shv(7, c) + SH_C2[4] * (xx - yy) * shv(8, c);
                if (D > 2) {
                    val = val + SH_C3[0] * Y * (3.0f * xx - yy) * shv(9, c) +
                          SH_C3[1] * xy * Z * shv(10, c) +
                          SH_C3[2] * Y * (4.0f * zz - xx - yy) * shv(11, c) +
                          SH_C3[3] * Z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * shv(12, c) +
                          SH_C3[4] * X * (4.0f * zz - xx - yy) * shv(13, c) +
                          SH_C3[5] * Z * (xx - yy) * shv(14, c) +
                          SH_C3[6] * X * (xx - 3.0f * yy) * shv(15, c);
                }
            }
        }
        val += 0.5f;
        if (val < 0.0f) flags |= (1u << c);
        res[c] = fmaxf(val, 0.0f);
        if (s.sh_jac) sh_direction_jacobian<D>(dx, dy, dz, [&](int k) { return shv(k, c); }, jac[3 * c], jac[3 * c + 1], jac[3 * c + 2]);
    }
    if (s.sh_jac) {  // for the per-splat backward of every pass of this camera (sh_jac.h)
        float4* jrec = reinterpret_cast<float4*>(s.sh_jac + ((size_t)v * st + i) * 12);
        jrec[0] = make_float4(jac[0], jac[1], jac[2], jac[3]);
        jrec[1] = make_float4(jac[4], jac[5], jac[6], jac[7]);
        jrec[2] = make_float4(jac[8], __uint_as_float(flags), 0.0f, 0.0f);  // + the clamp bits: the backward needs nothing else of this record
    }

    // Cull threshold: a pixel contributes only if alpha = min(0.99, opacity*exp(-q)) >= 1/255, i.e. only where the
    // conic's quadratic form q <= tau = ln(255*opacity).  The render kernels skip (splat, 8x8 block) pairs whose
    // minimum q over the block exceeds tau — pairs the blend would skip anyway.  1 % + 0.01 of margin covers the
    // fp32 error of conic / log / the block test; tau < 0 marks splats that can never reach 1/255; a conic that is
    // not positive definite (the block test assumes convexity) or culling switched off gives tau = 3e38.
    const float opacity = in.opacity;
    float hx, hy = 0.0f;
    {
        const float dc = conx * conz - cony * cony;
        if (!d.cull || !(dc > 0.0f && conx > 0.0f && conz > 0.0f)) hx = 3.0e38f;
        else if (!(opacity >= 0.0039f)) hx = -1.0f;  // 1/255 = 0.003921...: alpha <= opacity can never reach it
        else hx = fmaxf(0.0f, __logf(255.0f * opacity)) * 1.01f + 0.01f;
    }

    GeomRec g;
    g.x = pix_x; g.y = pix_y; g.conA = conx; g.conB = cony;
    g.conC = conz; g.opacity = opacity; g.r = res[0]; g.g = res[1];
    g.b = res[2]; g.hx = hx; g.hy = hy; g.depth = pvz;
    g.radius = (int)my_radius; g.flags = flags;
    g.rect_min = (uint32_t)rminx | ((uint32_t)rminy << 16);
    g.rect_max = (uint32_t)rmaxx | ((uint32_t)rmaxy << 16);
    *rec = g;
    const uint32_t ntiles = (uint32_t)((rmaxy - rminy) * (rmaxx - rminx));
    *tt = ntiles;

    // coarse binning: count the splat once per 64x64-px super-tile it touches, in the workgroup's LDS histogram.
    // (Global atomics for this cost 125 us of a 165 us launch: MI355X retires only ~10 G scattered atomics/s.)
    const int sx0 = rminx / STILE, sx1 = (rmaxx - 1) / STILE + 1, sy0 = rminy / STILE, sy1 = (rmaxy - 1) / STILE + 1;
    // (Dims::cut: a candidate behind the depth bound of every tile of the super-tile is not counted — k_coarse_scatter applies the same test)
    const uint32_t* zc = d.cut ? s.stile_zcut + (size_t)v * d.NST : nullptr;
    const uint32_t dbits = __float_as_uint(pvz);
    for (int sy = sy0; sy < sy1; sy++)
        for (int sx = sx0; sx < sx1; sx++)
            if (!zc || dbits <= zc[sy * d.sgx + sx]) atomicAdd(&hist[sy * d.sgx + sx], 1u);
    return ntiles;
}

#ifndef GS_PREPROCESS_CAMERAS
#define GS_PREPROCESS_CAMERAS 8  // cameras a block projects its splats through (tuning hook, tools/build_variant.sh)
#endif
// blockIdx.y selects a chunk of `cpb` geometry groups (cameras): the block loads its 256 splats once and projects them through
// every camera of the chunk, one LDS histogram per camera.
template <int D, bool H>
__global__ __launch_bounds__(WG) void k_preprocess(Dims d, const float* __restrict__ params, Scratch s, int cpb) {
    extern __shared__ uint32_t hist[];  // [cpb][NST] candidates of this block per (camera, super-tile)
    __shared__ uint32_t wsum[WG / 64];
    const int i = blockIdx.x * WG + threadIdx.x;
    const int v0 = blockIdx.y * cpb, v1 = min(d.VG, v0 + cpb);
    for (int k = threadIdx.x; k < (v1 - v0) * d.NST; k += WG) hist[k] = 0;
    SplatIn<D> in;
    if (i < d.P) load_splat<D, H>(d, params, s, i, in);
    if (s.mean_copy && blockIdx.y == 0 && i < d.P) {     // k_splat_bwd_reduce<D, true> (the fused update) reads the position from here
#pragma unroll
        for (int c = 0; c < 3; c++) s.mean_copy[(size_t)c * d.Pa + i] = in.loc[c];
    }
    __syncthreads();
    for (int v = v0; v < v1; v++) {
        uint32_t* h = hist + (size_t)(v - v0) * d.NST;
        uint32_t n = 0;
        if (i < d.P) n = preprocess_one<D>(d, in, s, i, v, h);
        // the block's share of the offsets scan (k_coarse_colscan's extra workgroup turns the block sums into prefixes, the
        // coarse scatter finishes the scan inside each block): no separate pass over tiles_touched
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) n += (uint32_t)__shfl_xor((int)n, o);
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = n;
        __syncthreads();  // also: every thread's LDS atomics into this camera's histogram are done
        if (threadIdx.x == 0) s.block_sums[(size_t)v * splat_blocks(d.Pa) + blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        // the block's row of the (block x super-tile) count matrix: the binning needs no global atomics at all
        uint32_t* row = s.wg_hist + ((size_t)v * splat_blocks(d.Pa) + blockIdx.x) * d.NST;
        for (int k = threadIdx.x; k < d.NST; k += WG) row[k] = h[k];
        __syncthreads();  // wsum is reused by the next camera
    }
}

int launch_preprocess(const Dims& d, const float* params, const Scratch& s, hipStream_t st) {
    if (d.P == 0 || d.VG == 0) return GS_OK;
    // cameras per block: as many as fit 32 KB of LDS histograms, at most GS_PREPROCESS_CAMERAS
    // (one camera at a time beyond 8192 super-tiles: 64 KB of counters at the 16384 of an 8192 x 8192 render)
    const int cpb = std::max(1, std::min({ d.VG, GS_PREPROCESS_CAMERAS, (int)(32768 / ((size_t)d.NST * sizeof(uint32_t))) }));
    dim3 grid((d.P + WG - 1) / WG, (d.VG + cpb - 1) / cpb);
    const size_t lds = (size_t)cpb * d.NST * sizeof(uint32_t);
    if (lds > 48 * 1024) {
        const void* k = nullptr;
        switch (d.D * 2 + (s.sh16 ? 1 : 0)) {
            case 0: k = (const void*)k_preprocess<0, false>; break;
            case 1: k = (const void*)k_preprocess<0, true>; break;
            case 2: k = (const void*)k_preprocess<1, false>; break;
            case 3: k = (const void*)k_preprocess<1, true>; break;
            case 4: k = (const void*)k_preprocess<2, false>; break;
            case 5: k = (const void*)k_preprocess<2, true>; break;
            case 6: k = (const void*)k_preprocess<3, false>; break;
            default: k = (const void*)k_preprocess<3, true>; break;
        }
        GS_TRY(allow_dynamic_lds(k, lds));
    }
    if (s.sh16) {
        switch (d.D) {
            case 0: hipLaunchKernelGGL((k_preprocess<0, true>), grid, dim3(WG), lds, st, d, params, s, cpb); break;
            case 1: hipLaunchKernelGGL((k_preprocess<1, true>), grid, dim3(WG), lds, st, d, params, s, cpb); break;
            case 2: hipLaunchKernelGGL((k_preprocess<2, true>), grid, dim3(WG), lds, st, d, params, s, cpb); break;
            default: hipLaunchKernelGGL((k_preprocess<3, true>), grid, dim3(WG), lds, st, d, params, s, cpb); break;
        }
    } else {
        switch (d.D) {
            case 0: hipLaunchKernelGGL((k_preprocess<0, false>), grid, dim3(WG), lds, st, d, params, s, cpb); break;
            case 1: hipLaunchKernelGGL((k_preprocess<1, false>), grid, dim3(WG), lds, st, d, params, s, cpb); break;
            case 2: hipLaunchKernelGGL((k_preprocess<2, false>), grid, dim3(WG), lds, st, d, params, s, cpb); break;
            default: hipLaunchKernelGGL((k_preprocess<3, false>), grid, dim3(WG), lds, st, d, params, s, cpb); break;
        }
    }
    GS_HIP(hipGetLastError());
    return GS_OK;
}

}  // namespace gs
