// k_update.hip — parameter update, layout transposes and the reference's two image kernels.
//   update      : applyGradients (src/Trainer.cu:81-101) as one launch over the SoA planes, rule
//                 GS_UPDATE_SGD_CLAMP = the reference; GS_UPDATE_ADAM = Adam ascent + the same clamps.
//   aos<->soa   : ModelSplatsHost layout (src/ModelSplatsHost.h:16-20) <-> plane-major device layout.
//   image ops   : imageFloatToInt (src/Trainer.cu:19-29), imageIntToLoss (src/Trainer.cu:33-44).
// All HBM-bound elementwise work: plane-major SoA makes every wave access 256 contiguous bytes.
#include <algorithm>

#include <hip/hip_fp16.h>

#include "gs_internal.h"

namespace gs {

// [lo, hi): the flat element range (plane * Pa + splat) this launch owns — everything on one GPU, the rank's chunk
// under the data-parallel sharded update; blockIdx.y counts planes from the first one the range touches.
__global__ __launch_bounds__(WG) void k_update(UpdateArgs u, int P, int Pa, float* __restrict__ params,
                                               const float* __restrict__ grads, float* __restrict__ am,
                                               float* __restrict__ av, size_t lo, size_t hi, int first_plane,
                                               __half* __restrict__ sh16) {
    const int i = blockIdx.x * WG + threadIdx.x;
    const int p = first_plane + blockIdx.y;
    if (i >= P) return;
    if ((size_t)p * Pa + i < lo || (size_t)p * Pa + i >= hi) return;
    const Planes pl{ u.M };
    float lr; int kind;
    update_plane_rule(u, pl, p, lr, kind);
    const size_t idx = (size_t)p * Pa + i;
    const float x = update_element(u, lr, kind, params[idx], grads[idx], am, av, idx);
    params[idx] = x;
    // trainer option "sh_fp16": the projection reads the SH coefficients from a half-precision copy (BASELINE cfg5);
    // the fp32 master above stays the optimiser's state — a learning rate of 1e-4 would vanish below half an fp16 ulp
    if (sh16 && p >= 3 && p < pl.scale(0)) sh16[(size_t)(p - 3) * Pa + i] = __float2half_rn(x);
}

// streaming-copy probe (gs_debug_hbm_copy_rate): float4 per lane, U independent loads in flight per trip, grid-stride; NT: non-temporal
// (streaming) loads and stores.  The caller times a few (grid, U, NT) forms and reports the fastest: the achievable ceiling, not one guess at it.
template <int U, bool NT>
__global__ __launch_bounds__(WG) void k_copy_probe(const float4* __restrict__ src, float4* __restrict__ dst, size_t n4) {
    const size_t stride = (size_t)gridDim.x * WG;
    size_t k = (size_t)blockIdx.x * WG + threadIdx.x;
    for (; k + (U - 1) * stride < n4; k += U * stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (NT) {
                const float* p = reinterpret_cast<const float*>(src + k + u * stride);
                v[u] = make_float4(__builtin_nontemporal_load(p), __builtin_nontemporal_load(p + 1), __builtin_nontemporal_load(p + 2), __builtin_nontemporal_load(p + 3));
            } else v[u] = src[k + u * stride];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (NT) {
                float* p = reinterpret_cast<float*>(dst + k + u * stride);
                __builtin_nontemporal_store(v[u].x, p); __builtin_nontemporal_store(v[u].y, p + 1); __builtin_nontemporal_store(v[u].z, p + 2); __builtin_nontemporal_store(v[u].w, p + 3);
            } else dst[k + u * stride] = v[u];
        }
    }
    for (; k < n4; k += stride) dst[k] = src[k];
}
int launch_copy_probe(const void* src, void* dst, size_t bytes, int form, hipStream_t st) {
    const size_t n4 = bytes / 16;
    if (!n4) return GS_OK;
    // form: bits 0-1 grid (256 x 8 / 16 / 32 / 64 workgroups), bit 2 unroll 4 (else 1), bit 3 non-temporal
    const unsigned grid = (unsigned)std::min<size_t>((n4 + WG - 1) / WG, (size_t)256 * (8u << (form & 3)));
    const float4* a = (const float4*)src;
    float4* b = (float4*)dst;
    switch ((form >> 2) & 3) {
        case 0: hipLaunchKernelGGL((k_copy_probe<1, false>), dim3(grid), dim3(WG), 0, st, a, b, n4); break;
        case 1: hipLaunchKernelGGL((k_copy_probe<4, false>), dim3(grid), dim3(WG), 0, st, a, b, n4); break;
        case 2: hipLaunchKernelGGL((k_copy_probe<1, true>), dim3(grid), dim3(WG), 0, st, a, b, n4); break;
        default: hipLaunchKernelGGL((k_copy_probe<4, true>), dim3(grid), dim3(WG), 0, st, a, b, n4); break;
    }
    GS_HIP(hipGetLastError());
    return GS_OK;
}

__global__ __launch_bounds__(WG) void k_sh_to_half(int P, int Pa, const float* __restrict__ planes, __half* __restrict__ sh16) {
    const int i = blockIdx.x * WG + threadIdx.x;
    if (i >= P) return;
    const size_t k = blockIdx.y;
    sh16[k * Pa + i] = __float2half_rn(planes[(3 + k) * Pa + i]);
}
int launch_sh_to_half(int M, int P, int Pa, const float* planes, uint16_t* sh16, hipStream_t st) {
    if (P == 0) return GS_OK;
    hipLaunchKernelGGL(k_sh_to_half, dim3((P + WG - 1) / WG, 3 * M), dim3(WG), 0, st, P, Pa, planes, reinterpret_cast<__half*>(sh16));
    GS_HIP(hipGetLastError());
    return GS_OK;
}

UpdateArgs make_update_args(const Planes& pl, const gs_hyper& h, int adam_t) {
    UpdateArgs u;
    u.lr_loc = h.lr_location; u.lr_sh = h.lr_sh; u.lr_scale = h.lr_scale; u.lr_opac = h.lr_opacity; u.lr_rot = h.lr_rotation;
    u.scale_max = h.scale_max; u.rule = h.update_rule; u.b1 = h.adam_beta1; u.b2 = h.adam_beta2; u.eps = h.adam_eps;
    u.bc1 = 1.0f - powf(h.adam_beta1, (float)adam_t);
    u.bc2 = 1.0f - powf(h.adam_beta2, (float)adam_t);
    u.M = pl.M;
    return u;
}

int launch_update(const Planes& pl, int P, int Pa, float* params, float* grads, float* adam_m, float* adam_v, int adam_t,
                  const gs_hyper& h, hipStream_t st, size_t lo, size_t hi, uint16_t* sh16) {
    if (P == 0) return GS_OK;
    hi = std::min(hi, (size_t)pl.count() * Pa);
    if (lo >= hi) return GS_OK;
    const int first_plane = (int)(lo / Pa), last_plane = (int)((hi - 1) / Pa);
    const UpdateArgs u = make_update_args(pl, h, adam_t);
    hipLaunchKernelGGL(k_update, dim3((P + WG - 1) / WG, last_plane - first_plane + 1), dim3(WG), 0, st, u, P, Pa, params,
                       (const float*)grads, adam_m, adam_v, lo, hi, first_plane, reinterpret_cast<__half*>(sh16));
    GS_HIP(hipGetLastError());
    return GS_OK;
}

__global__ __launch_bounds__(WG) void k_aos_to_soa(int P, int Pa, int M, const float* __restrict__ loc,
                                                   const float* __restrict__ sh, const float* __restrict__ scale,
                                                   const float* __restrict__ opac, const float* __restrict__ rot,
                                                   float* __restrict__ planes) {
    const int i = blockIdx.x * WG + threadIdx.x;
    if (i >= P) return;
    const Planes pl{ M };
    const size_t st = (size_t)Pa;
    for (int c = 0; c < 3; c++) {
        planes[pl.loc(c) * st + i] = loc[3 * (size_t)i + c];
        planes[pl.scale(c) * st + i] = scale[3 * (size_t)i + c];
    }
    for (int k = 0; k < 3 * M; k++) planes[(3 + k) * st + i] = sh[(size_t)i * 3 * M + k];
    planes[pl.opac() * st + i] = opac[i];
    for (int c = 0; c < 4; c++) planes[pl.rot(c) * st + i] = rot[4 * (size_t)i + c];
}

__global__ __launch_bounds__(WG) void k_soa_to_aos(int P, int Pa, int M, const float* __restrict__ planes,
                                                   float* __restrict__ loc, float* __restrict__ sh,
                                                   float* __restrict__ scale, float* __restrict__ opac,
                                                   float* __restrict__ rot) {
    const int i = blockIdx.x * WG + threadIdx.x;
    if (i >= P) return;
    const Planes pl{ M };
    const size_t st = (size_t)Pa;
    for (int c = 0; c < 3; c++) {
        loc[3 * (size_t)i + c] = planes[pl.loc(c) * st + i];
        scale[3 * (size_t)i + c] = planes[pl.scale(c) * st + i];
    }
    for (int k = 0; k < 3 * M; k++) sh[(size_t)i * 3 * M + k] = planes[(3 + k) * st + i];
    opac[i] = planes[pl.opac() * st + i];
    for (int c = 0; c < 4; c++) rot[4 * (size_t)i + c] = planes[pl.rot(c) * st + i];
}

int launch_aos_to_soa(int P, int Pa, int M, const float* loc, const float* sh, const float* scale, const float* opac,
                      const float* rot, float* planes, hipStream_t st) {
    if (P == 0) return GS_OK;
    hipLaunchKernelGGL(k_aos_to_soa, dim3((P + WG - 1) / WG), dim3(WG), 0, st, P, Pa, M, loc, sh, scale, opac, rot, planes);
    GS_HIP(hipGetLastError());
    return GS_OK;
}
int launch_soa_to_aos(int P, int Pa, int M, const float* planes, float* loc, float* sh, float* scale, float* opac,
                      float* rot, hipStream_t st) {
    if (P == 0) return GS_OK;
    hipLaunchKernelGGL(k_soa_to_aos, dim3((P + WG - 1) / WG), dim3(WG), 0, st, P, Pa, M, planes, loc, sh, scale, opac, rot);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

// imageFloatToInt, src/Trainer.cu:19-29: x256 (not x255), clamp to [0,255], opaque alpha
__global__ __launch_bounds__(WG) void k_image_float_to_int(const float* __restrict__ src, uint32_t* __restrict__ fb, int n) {
    const int i = blockIdx.x * WG + threadIdx.x;
    if (i >= n) return;
    const uint32_t r = (uint32_t)min(255, max(0, (int)(src[i] * 256.0f)));
    const uint32_t g = (uint32_t)min(255, max(0, (int)(src[i + (size_t)n] * 256.0f)));
    const uint32_t b = (uint32_t)min(255, max(0, (int)(src[i + 2 * (size_t)n] * 256.0f)));
    fb[i] = (r << 0) + (g << 8) + (b << 16) + (0xFFu << 24);
}
// imageIntToLoss, src/Trainer.cu:33-44
__global__ __launch_bounds__(WG) void k_image_int_to_loss(const uint32_t* __restrict__ truth, const float* __restrict__ rast,
                                                          float* __restrict__ loss, int n) {
    const int i = blockIdx.x * WG + threadIdx.x;
    if (i >= n) return;
    const uint32_t t = truth[i];
    loss[i] = ((float)(t & 0xFF) / 255.0f) - rast[i];
    loss[i + (size_t)n] = ((float)((t >> 8) & 0xFF) / 255.0f) - rast[i + (size_t)n];
    loss[i + 2 * (size_t)n] = ((float)((t >> 16) & 0xFF) / 255.0f) - rast[i + 2 * (size_t)n];
}

int launch_image_float_to_int(const float* src, uint32_t* fb, int w, int h, hipStream_t st) {
    const int n = w * h;
    if (n == 0) return GS_OK;
    hipLaunchKernelGGL(k_image_float_to_int, dim3((n + WG - 1) / WG), dim3(WG), 0, st, src, fb, n);
    GS_HIP(hipGetLastError());
    return GS_OK;
}
int launch_image_int_to_loss(const uint32_t* truth, const float* rast, float* loss, int w, int h, hipStream_t st) {
    const int n = w * h;
    if (n == 0) return GS_OK;
    hipLaunchKernelGGL(k_image_int_to_loss, dim3((n + WG - 1) / WG), dim3(WG), 0, st, truth, rast, loss, n);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

}  // namespace gs
