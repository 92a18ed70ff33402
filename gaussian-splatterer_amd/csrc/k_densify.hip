// k_densify.hip — split / clone / prune on the device, directly on the SoA planes.
//
// The reference does this on the CPU every 200th iteration (src/Trainer.cu:433-542: download the model, walk
// std::unordered_sets, upload); at 100k splats that round trip costs 60 ms, i.e. 0.3 ms per iteration amortised —
// 12 % of a 2.4 ms step.  The same decisions are order-independent once the candidates are visited in ascending
// index (the reference's set order is implementation-defined; the oracle's restatement uses ascending index too),
// so they become three prefix sums and one emit pass, bit-identical to that restatement (oracle/gs_oracle.cpp, orc_densify):
//   cull   : opacity <= cullOpacity  or  |scale| < cullSize                                              (:451)
//   densify: var - |avgGradLoc| > densifyVariance; split if |scale| > splitSize else clone               (:453-454)
//   capacity: the s-th split (ascending index) happens iff original + s < capacity, then the c-th clone iff
//             original + splits_done + c < capacity                                                       (:458, :498)
//   split  : both copies move +-0.5*splitDistance along the largest scale axis rotated by q, scale *= splitScale,
//            the quaternion is stored back in glm member order                                            (:459-496)
//   clone  : the copy is offset by (R*scale) (.) normalize(avgGradLoc) * cloneDistance                    (:499-521)
//   prune  : stable compaction: kept originals in index order, then the split twins, then the clone twins (:524-534)
// Built with -ffp-contract=off like the rest of the library: every float operation is the individually rounded
// IEEE operation the restatement performs, in the same order.
#include "gs_internal.h"

namespace gs {

namespace {
enum : uint32_t { KEEP = 0, SPLIT = 1, CLONE = 2, REMOVE = 3 };

__device__ inline float norm3(float a, float b, float c) { return sqrtf(a * a + b * b + c * c); }

// (mat4)q * vec4(v, 1) followed by the reference's divide by w, as the reference's glm expression does (src/Trainer.cu:466-470)
__device__ inline void rotate(float w_, float x, float y, float z, const float v[3], float out[3]) {
    float R[3][3];  // R[col][row]
    R[0][0] = 1.0f - 2.0f * (y * y + z * z); R[0][1] = 2.0f * (x * y + w_ * z); R[0][2] = 2.0f * (x * z - w_ * y);
    R[1][0] = 2.0f * (x * y - w_ * z); R[1][1] = 1.0f - 2.0f * (x * x + z * z); R[1][2] = 2.0f * (y * z + w_ * x);
    R[2][0] = 2.0f * (x * z + w_ * y); R[2][1] = 2.0f * (y * z - w_ * x); R[2][2] = 1.0f - 2.0f * (x * x + y * y);
    const float w = 0.0f * v[0] + 0.0f * v[1] + 0.0f * v[2] + 1.0f * 1.0f;
#pragma unroll
    for (int r = 0; r < 3; r++) out[r] = (R[0][r] * v[0] + R[1][r] * v[1] + R[2][r] * v[2] + 0.0f * 1.0f) / w;
}
}  // namespace

// flags[0..2][i] = is split / is clone / is kept (u32 each, stride `fs`), action[i] for the emit pass
__global__ __launch_bounds__(WG) void k_densify_classify(int count, int Pa, int M, const float* __restrict__ params,
                                                         const float* __restrict__ grad, gs_hyper h, uint32_t* __restrict__ flags,
                                                         int fs) {
    const int i = blockIdx.x * WG + threadIdx.x;
    if (i >= count) return;
    const Planes pl{ M };
    const size_t st = (size_t)Pa;
    const float size = norm3(params[pl.scale(0) * st + i], params[pl.scale(1) * st + i], params[pl.scale(2) * st + i]);
    const float op = params[pl.opac() * st + i];
    uint32_t a = KEEP;
    if (op <= h.cull_opacity || size < h.cull_size) a = REMOVE;
    else {
        const float g = norm3(grad[pl.loc(0) * st + i], grad[pl.loc(1) * st + i], grad[pl.loc(2) * st + i]);
        if (grad[pl.var() * st + i] - g > h.densify_variance) a = size > h.split_size ? SPLIT : CLONE;
    }
    flags[i] = a == SPLIT;
    flags[(size_t)fs + i] = a == CLONE;
    flags[2 * (size_t)fs + i] = a != REMOVE;
}

// ranks = inclusive scans of the three flag arrays
__global__ __launch_bounds__(WG) void k_densify_emit(int count, int Pa, int M, const float* __restrict__ params,
                                                     const float* __restrict__ grad, gs_hyper h, const uint32_t* __restrict__ flags,
                                                     const uint32_t* __restrict__ ranks, int fs, int splits_done, int clones_done,
                                                     int kept, int outPa, float* __restrict__ out) {
    const int i = blockIdx.x * WG + threadIdx.x;
    if (i >= count) return;
    if (!flags[2 * (size_t)fs + i]) return;  // removed
    const Planes pl{ M };
    const size_t st = (size_t)Pa, so = (size_t)outPa;
    const bool is_split = flags[i] != 0u, is_clone = flags[(size_t)fs + i] != 0u;
    const int dest = (int)ranks[2 * (size_t)fs + i] - 1;
    float loc[3], sc[3], q[4];
#pragma unroll
    for (int c = 0; c < 3; c++) { loc[c] = params[pl.loc(c) * st + i]; sc[c] = params[pl.scale(c) * st + i]; }
#pragma unroll
    for (int c = 0; c < 4; c++) q[c] = params[pl.rot(c) * st + i];
    const float op = params[pl.opac() * st + i];
    int twin = -1;
    float tloc[3] = { loc[0], loc[1], loc[2] };
    const bool do_split = is_split && (int)ranks[i] - 1 < splits_done;
    const bool do_clone = is_clone && (int)ranks[(size_t)fs + i] - 1 < clones_done;
    if (do_split) {
        float axis[3] = { sc[0], sc[1], sc[2] };
        if (sc[0] > sc[1] && sc[0] > sc[2]) { axis[1] *= 0.0f; axis[2] *= 0.0f; }
        else if (sc[1] > sc[2]) { axis[0] *= 0.0f; axis[2] *= 0.0f; }
        else { axis[0] *= 0.0f; axis[1] *= 0.0f; }
        float off[3];
        rotate(q[0], q[1], q[2], q[3], axis, off);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float centre = loc[k];
            loc[k] = centre + off[k] * h.split_distance * 0.5f;
            tloc[k] = centre - off[k] * h.split_distance * 0.5f;
            sc[k] = sc[k] * h.split_scale;
        }
        if (h.quat_layout == GS_QUAT_XYZW) { const float w = q[0]; q[0] = q[1]; q[1] = q[2]; q[2] = q[3]; q[3] = w; }
        twin = kept + (int)ranks[i] - 1;
    } else if (do_clone) {
        const float g[3] = { grad[pl.loc(0) * st + i], grad[pl.loc(1) * st + i], grad[pl.loc(2) * st + i] };
        const float inv = 1.0f / norm3(g[0], g[1], g[2]);  // glm::normalize = v * inversesqrt(dot(v,v))
        float off[3];
        rotate(q[0], q[1], q[2], q[3], sc, off);
#pragma unroll
        for (int k = 0; k < 3; k++) tloc[k] = loc[k] + off[k] * (g[k] * inv) * h.clone_distance;
        twin = kept + splits_done + (int)ranks[(size_t)fs + i] - 1;
    }
#pragma unroll
    for (int c = 0; c < 3; c++) { out[pl.loc(c) * so + dest] = loc[c]; out[pl.scale(c) * so + dest] = sc[c]; }
#pragma unroll
    for (int c = 0; c < 4; c++) out[pl.rot(c) * so + dest] = q[c];
    out[pl.opac() * so + dest] = op;
    if (twin >= 0) {
#pragma unroll
        for (int c = 0; c < 3; c++) { out[pl.loc(c) * so + twin] = tloc[c]; out[pl.scale(c) * so + twin] = sc[c]; }
#pragma unroll
        for (int c = 0; c < 4; c++) out[pl.rot(c) * so + twin] = q[c];
        out[pl.opac() * so + twin] = op;
    }
    for (int k = 0; k < 3 * M; k++) {
        const float v = params[(size_t)(3 + k) * st + i];
        out[(size_t)(3 + k) * so + dest] = v;
        if (twin >= 0) out[(size_t)(3 + k) * so + twin] = v;
    }
}

// Optimizer state through densify (build-side extension; the reference has no optimizer state): the Adam moments of
// a kept splat move to its new index, both halves of a split and a clone's twin inherit the parent's moments, and the
// rotation rows follow the quaternion's member permutation of a split (k_densify_emit).  Same index bookkeeping as the
// emit pass; blockIdx.y selects the first or the second moment buffer.
__global__ __launch_bounds__(WG) void k_densify_carry(int count, int Pa, int M, gs_hyper h, const uint32_t* __restrict__ flags,
                                                      const uint32_t* __restrict__ ranks, int fs, int splits_done, int clones_done, int kept,
                                                      int outPa, const float* __restrict__ src_m, float* __restrict__ dst_m,
                                                      const float* __restrict__ src_v, float* __restrict__ dst_v) {
    const int i = blockIdx.x * WG + threadIdx.x;
    if (i >= count) return;
    if (!flags[2 * (size_t)fs + i]) return;
    const float* __restrict__ src = blockIdx.y ? src_v : src_m;
    float* __restrict__ dst = blockIdx.y ? dst_v : dst_m;
    const Planes pl{ M };
    const size_t st = (size_t)Pa, so = (size_t)outPa;
    const int dest = (int)ranks[2 * (size_t)fs + i] - 1;
    const bool do_split = flags[i] != 0u && (int)ranks[i] - 1 < splits_done;
    const bool do_clone = flags[(size_t)fs + i] != 0u && (int)ranks[(size_t)fs + i] - 1 < clones_done;
    const int twin = do_split ? kept + (int)ranks[i] - 1 : (do_clone ? kept + splits_done + (int)ranks[(size_t)fs + i] - 1 : -1);
    const bool permute = do_split && h.quat_layout == GS_QUAT_XYZW;
    const int nplanes = pl.count();
    for (int p = 0; p < nplanes; p++) {
        int ps = p;
        if (permute && p >= pl.rot(0)) ps = pl.rot(((p - pl.rot(0)) + 1) & 3);  // stored (q1, q2, q3, q0): row c takes old row c + 1
        const float v = src[(size_t)ps * st + i];
        dst[(size_t)p * so + dest] = v;
        if (twin >= 0) dst[(size_t)p * so + twin] = v;
    }
}

int launch_densify_carry(int count, int Pa, int M, const gs_hyper& h, const uint32_t* flags, const uint32_t* ranks, int fs, int splits_done,
                         int clones_done, int kept, int outPa, const float* src_m, float* dst_m, const float* src_v, float* dst_v,
                         hipStream_t st) {
    if (count == 0) return GS_OK;
    hipLaunchKernelGGL(k_densify_carry, dim3((count + WG - 1) / WG, 2), dim3(WG), 0, st, count, Pa, M, h, flags, ranks, fs, splits_done,
                       clones_done, kept, outPa, src_m, dst_m, src_v, dst_v);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

// Launch helpers.  `flags` and `ranks` are 3 x fs u32 each; `partials` as launch_scan_u32 needs for (count, batch 3).
int launch_densify_classify(int count, int Pa, int M, const float* params, const float* grad, const gs_hyper& h, uint32_t* flags,
                            uint32_t* ranks, int fs, uint32_t* partials, hipStream_t st) {
    if (count == 0) return GS_OK;
    hipLaunchKernelGGL(k_densify_classify, dim3((count + WG - 1) / WG), dim3(WG), 0, st, count, Pa, M, params, grad, h, flags, fs);
    GS_HIP(hipGetLastError());
    GS_TRY(launch_scan_u32(flags, ranks, count, fs, 3, partials, st));
    return GS_OK;
}

int launch_densify_emit(int count, int Pa, int M, const float* params, const float* grad, const gs_hyper& h, const uint32_t* flags,
                        const uint32_t* ranks, int fs, int splits_done, int clones_done, int kept, int outPa, float* out,
                        hipStream_t st) {
    if (count == 0) return GS_OK;
    hipLaunchKernelGGL(k_densify_emit, dim3((count + WG - 1) / WG), dim3(WG), 0, st, count, Pa, M, params, grad, h, flags, ranks, fs,
                       splits_done, clones_done, kept, outPa, out);
    GS_HIP(hipGetLastError());
    return GS_OK;
}

}  // namespace gs
