// k_splat_bwd.hip — per-splat backward: conic -> cov2D -> cov3D & mean (EWA Jacobian), mean2D ->
// mean (projection), colour -> SH & mean (view direction), cov3D -> scale & rotation.
// Replaces computeCov2DCUDA + preprocessCUDA inside CudaRasterizer::Rasterizer::backward
// (reference call site src/Trainer.cu:378-412) and, in the trainer
// form, fuses the reference's nine cudaMemsets (src/Trainer.cu:366-375) and accumulateGradients
// (src/Trainer.cu:47-77): one thread owns one splat, walks the views in the reference's order,
// sums that splat's (splat,tile) gradient rows (contiguous slots, fixed order), and writes every
// averaged-gradient plane exactly once per step — no per-view read-modify-write, no atomics.
#include <hip/hip_fp16.h>

#include "gs_internal.h"
#include "sh_jac.h"

namespace gs {

namespace {
constexpr float C0 = 0.28209479177387814f;
constexpr float C1 = 0.4886025119029199f;
constexpr float C2_0 = 1.0925484305920792f, C2_1 = -1.0925484305920792f, C2_2 = 0.31539156525252005f,
                C2_3 = -1.0925484305920792f, C2_4 = 0.5462742152960396f;
constexpr float C3_0 = -0.5900435899266435f, C3_1 = 2.890611442640554f, C3_2 = -0.4570457994644658f,
                C3_3 = 0.3731763325901154f, C3_4 = -0.4570457994644658f, C3_5 = 1.445305721320277f,
                C3_6 = -0.5900435899266435f;
}  // namespace

template <int D> struct SplatOut {
    static constexpr int NC = (D + 1) * (D + 1);
    float mean[3], scale[3], rot[4], cov3D[6];
    float sh[NC][3];
};

// in: mean, s = mod*scale, quaternion (r,x,y,z), view, clamp flags, the pixel-stage sums (colour 3, mean2D 2, conic
// x/y/w 3) and `jac(ch, X, Y, Z, dx, dy, dz)` = d colour[ch] / d direction (sh_jac.h: evaluated from the SH
// coefficients, or read back from the projection's record).  Mirrors oracle/gs_oracle.cpp::preprocess_backward.
template <int D, bool WANT_SH, class JacAt>
__device__ inline void splat_backward_core(const gs_view& vp, int W, int H, const float mean[3], const float s[3],
                                           const float q[4], JacAt jac, uint32_t clamp_flags,
                                           const float dcolor[3], float g2x, float g2y, float gcx, float gcy, float gcz,
                                           SplatOut<D>& o, float dRGB_out[3]) {
    const float* view = vp.view;
    const float* proj = vp.projview;
    const float focal_x = (float)W / (2.0f * vp.tan_fovx);
    const float focal_y = (float)H / (2.0f * vp.tan_fovy);
    const float r = q[0], x = q[1], y = q[2], z = q[3];
    float Rg[3][3];
    Rg[0][0] = 1.0f - 2.0f * (y * y + z * z); Rg[0][1] = 2.0f * (x * y - r * z); Rg[0][2] = 2.0f * (x * z + r * y);
    Rg[1][0] = 2.0f * (x * y + r * z); Rg[1][1] = 1.0f - 2.0f * (x * x + z * z); Rg[1][2] = 2.0f * (y * z - r * x);
    Rg[2][0] = 2.0f * (x * z - r * y); Rg[2][1] = 2.0f * (y * z + r * x); Rg[2][2] = 1.0f - 2.0f * (x * x + y * y);
    float Mm[3][3];
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int k = 0; k < 3; k++) Mm[c][k] = s[k] * Rg[c][k];
#define GS_SIG(c, rr) (Mm[rr][0] * Mm[c][0] + Mm[rr][1] * Mm[c][1] + Mm[rr][2] * Mm[c][2])
    const float c3[6] = { GS_SIG(0, 0), GS_SIG(0, 1), GS_SIG(0, 2), GS_SIG(1, 1), GS_SIG(1, 2), GS_SIG(2, 2) };
#undef GS_SIG

    // ---- cov2D backward ----
    float tx = view[0] * mean[0] + view[4] * mean[1] + view[8] * mean[2] + view[12];
    float ty = view[1] * mean[0] + view[5] * mean[1] + view[9] * mean[2] + view[13];
    const float tz_ = view[2] * mean[0] + view[6] * mean[1] + view[10] * mean[2] + view[14];
    const float limx = 1.3f * vp.tan_fovx, limy = 1.3f * vp.tan_fovy;
    const float txtz = tx / tz_, tytz = ty / tz_;
    tx = fminf(limx, fmaxf(-limx, txtz)) * tz_;
    ty = fminf(limy, fmaxf(-limy, tytz)) * tz_;
    const float x_grad_mul = (txtz < -limx || txtz > limx) ? 0.0f : 1.0f;
    const float y_grad_mul = (tytz < -limy || tytz > limy) ? 0.0f : 1.0f;
    const float J00 = focal_x / tz_, J02 = -(focal_x * tx) / (tz_ * tz_);
    const float J11 = focal_y / tz_, J12 = -(focal_y * ty) / (tz_ * tz_);
    float T[2][3];
#pragma unroll
    for (int rr = 0; rr < 3; rr++) {
        T[0][rr] = view[4 * rr] * J00 + view[4 * rr + 2] * J02;
        T[1][rr] = view[4 * rr + 1] * J11 + view[4 * rr + 2] * J12;
    }
    const float V[3][3] = { { c3[0], c3[1], c3[2] }, { c3[1], c3[3], c3[4] }, { c3[2], c3[4], c3[5] } };
    float A[3][2];
#pragma unroll
    for (int k = 0; k < 3; k++)
#pragma unroll
        for (int rr = 0; rr < 2; rr++) A[k][rr] = T[rr][0] * V[0][k] + T[rr][1] * V[1][k] + T[rr][2] * V[2][k];
    const float a = A[0][0] * T[0][0] + A[1][0] * T[0][1] + A[2][0] * T[0][2] + 0.3f;
    const float b = A[0][1] * T[0][0] + A[1][1] * T[0][1] + A[2][1] * T[0][2];
    const float c = A[0][1] * T[1][0] + A[1][1] * T[1][1] + A[2][1] * T[1][2] + 0.3f;
    const float denom = a * c - b * b;
    float dL_da = 0, dL_db = 0, dL_dc = 0;
    const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
    float* dcov = o.cov3D;
    if (denom2inv != 0.0f) {
        dL_da = denom2inv * (-c * c * gcx + 2.0f * b * c * gcy + (denom - a * c) * gcz);
        dL_dc = denom2inv * (-a * a * gcz + 2.0f * a * b * gcy + (denom - a * c) * gcx);
        dL_db = denom2inv * 2.0f * (b * c * gcx - (denom + 2.0f * b * b) * gcy + a * b * gcz);
        dcov[0] = T[0][0] * T[0][0] * dL_da + T[0][0] * T[1][0] * dL_db + T[1][0] * T[1][0] * dL_dc;
        dcov[3] = T[0][1] * T[0][1] * dL_da + T[0][1] * T[1][1] * dL_db + T[1][1] * T[1][1] * dL_dc;
        dcov[5] = T[0][2] * T[0][2] * dL_da + T[0][2] * T[1][2] * dL_db + T[1][2] * T[1][2] * dL_dc;
        dcov[1] = 2.0f * T[0][0] * T[0][1] * dL_da + (T[0][0] * T[1][1] + T[0][1] * T[1][0]) * dL_db + 2.0f * T[1][0] * T[1][1] * dL_dc;
        dcov[2] = 2.0f * T[0][0] * T[0][2] * dL_da + (T[0][0] * T[1][2] + T[0][2] * T[1][0]) * dL_db + 2.0f * T[1][0] * T[1][2] * dL_dc;
        dcov[4] = 2.0f * T[0][2] * T[0][1] * dL_da + (T[0][1] * T[1][2] + T[0][2] * T[1][1]) * dL_db + 2.0f * T[1][1] * T[1][2] * dL_dc;
    } else {
#pragma unroll
        for (int k = 0; k < 6; k++) dcov[k] = 0.0f;
    }
    float dT[2][3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float tv0 = T[0][0] * V[k][0] + T[0][1] * V[k][1] + T[0][2] * V[k][2];
        const float tv1 = T[1][0] * V[k][0] + T[1][1] * V[k][1] + T[1][2] * V[k][2];
        dT[0][k] = 2.0f * tv0 * dL_da + tv1 * dL_db;
        dT[1][k] = 2.0f * tv1 * dL_dc + tv0 * dL_db;
    }
    const float dJ00 = view[0] * dT[0][0] + view[4] * dT[0][1] + view[8] * dT[0][2];
    const float dJ02 = view[2] * dT[0][0] + view[6] * dT[0][1] + view[10] * dT[0][2];
    const float dJ11 = view[1] * dT[1][0] + view[5] * dT[1][1] + view[9] * dT[1][2];
    const float dJ12 = view[2] * dT[1][0] + view[6] * dT[1][1] + view[10] * dT[1][2];
    const float tz = 1.0f / tz_, tz2 = tz * tz, tz3 = tz2 * tz;
    const float dtx = x_grad_mul * -focal_x * tz2 * dJ02;
    const float dty = y_grad_mul * -focal_y * tz2 * dJ12;
    const float dtz = -focal_x * tz2 * dJ00 - focal_y * tz2 * dJ11 + (2.0f * focal_x * tx) * tz3 * dJ02 +
                      (2.0f * focal_y * ty) * tz3 * dJ12;
    float dmx = view[0] * dtx + view[1] * dty + view[2] * dtz;
    float dmy = view[4] * dtx + view[5] * dty + view[6] * dtz;
    float dmz = view[8] * dtx + view[9] * dty + view[10] * dtz;

    // ---- projection of the mean2D gradient ----
    {
        const float mhw = proj[3] * mean[0] + proj[7] * mean[1] + proj[11] * mean[2] + proj[15];
        const float m_w = 1.0f / (mhw + 0.0000001f);
        const float mul1 = (proj[0] * mean[0] + proj[4] * mean[1] + proj[8] * mean[2] + proj[12]) * m_w * m_w;
        const float mul2 = (proj[1] * mean[0] + proj[5] * mean[1] + proj[9] * mean[2] + proj[13]) * m_w * m_w;
        dmx += (proj[0] * m_w - proj[3] * mul1) * g2x + (proj[1] * m_w - proj[3] * mul2) * g2y;
        dmy += (proj[4] * m_w - proj[7] * mul1) * g2x + (proj[5] * m_w - proj[7] * mul2) * g2y;
        dmz += (proj[8] * m_w - proj[11] * mul1) * g2x + (proj[9] * m_w - proj[11] * mul2) * g2y;
    }

    // ---- SH backward ----
    {
        const float ox = mean[0] - vp.campos[0], oy = mean[1] - vp.campos[1], oz = mean[2] - vp.campos[2];
        const float len = sqrtf(ox * ox + oy * oy + oz * oz);
        const float X = ox / len, Y = oy / len, Z = oz / len;
        float ddx = 0, ddy = 0, ddz = 0;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            const float dl = dcolor[ch] * (((clamp_flags >> ch) & 1u) ? 0.0f : 1.0f);
            float dx_ = 0, dy_ = 0, dz_ = 0;
            dRGB_out[ch] = dl;
            if constexpr (WANT_SH) o.sh[0][ch] = C0 * dl;
            if constexpr (WANT_SH && D > 0) { o.sh[1][ch] = (-C1 * Y) * dl; o.sh[2][ch] = (C1 * Z) * dl; o.sh[3][ch] = (-C1 * X) * dl; }
            if constexpr (WANT_SH && D > 1) {
                const float xx = X * X, yy = Y * Y, zz = Z * Z, xy = X * Y, yz = Y * Z, xz = X * Z;
                o.sh[4][ch] = (C2_0 * xy) * dl; o.sh[5][ch] = (C2_1 * yz) * dl;
                o.sh[6][ch] = (C2_2 * (2.0f * zz - xx - yy)) * dl; o.sh[7][ch] = (C2_3 * xz) * dl;
                o.sh[8][ch] = (C2_4 * (xx - yy)) * dl;
                if constexpr (D > 2) {
                    o.sh[9][ch] = (C3_0 * Y * (3.0f * xx - yy)) * dl;
                    o.sh[10][ch] = (C3_1 * xy * Z) * dl;
                    o.sh[11][ch] = (C3_2 * Y * (4.0f * zz - xx - yy)) * dl;
                    o.sh[12][ch] = (C3_3 * Z * (2.0f * zz - 3.0f * xx - 3.0f * yy)) * dl;
                    o.sh[13][ch] = (C3_4 * X * (4.0f * zz - xx - yy)) * dl;
                    o.sh[14][ch] = (C3_5 * Z * (xx - yy)) * dl;
                    o.sh[15][ch] = (C3_6 * X * (xx - 3.0f * yy)) * dl;
                }
            }
            jac(ch, X, Y, Z, dx_, dy_, dz_);
            ddx += dx_ * dl; ddy += dy_ * dl; ddz += dz_ * dl;
        }
        const float sum2 = ox * ox + oy * oy + oz * oz;
        const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
        dmx += ((+sum2 - ox * ox) * ddx - oy * ox * ddy - oz * ox * ddz) * invsum32;
        dmy += (-ox * oy * ddx + (sum2 - oy * oy) * ddy - oz * oy * ddz) * invsum32;
        dmz += (-ox * oz * ddx - oy * oz * ddy + (sum2 - oz * oz) * ddz) * invsum32;
    }
    o.mean[0] = dmx; o.mean[1] = dmy; o.mean[2] = dmz;

    // ---- cov3D -> scale, rotation ----
    {
        const float dS[3][3] = { { dcov[0], 0.5f * dcov[1], 0.5f * dcov[2] },
                                 { 0.5f * dcov[1], dcov[3], 0.5f * dcov[4] },
                                 { 0.5f * dcov[2], 0.5f * dcov[4], dcov[5] } };
        float dMt[3][3];  // dMt[k][c] = dM[c][k],  dM[c][k] = 2 * sum_j M[j][k] * dS[c][j]
#pragma unroll
        for (int cc = 0; cc < 3; cc++)
#pragma unroll
            for (int k = 0; k < 3; k++) dMt[k][cc] = 2.0f * (Mm[0][k] * dS[cc][0] + Mm[1][k] * dS[cc][1] + Mm[2][k] * dS[cc][2]);
#pragma unroll
        for (int k = 0; k < 3; k++) o.scale[k] = Rg[0][k] * dMt[k][0] + Rg[1][k] * dMt[k][1] + Rg[2][k] * dMt[k][2];
#pragma unroll
        for (int k = 0; k < 3; k++)
#pragma unroll
            for (int cc = 0; cc < 3; cc++) dMt[k][cc] *= s[k];
        o.rot[0] = 2.0f * z * (dMt[0][1] - dMt[1][0]) + 2.0f * y * (dMt[2][0] - dMt[0][2]) + 2.0f * x * (dMt[1][2] - dMt[2][1]);
        o.rot[1] = 2.0f * y * (dMt[1][0] + dMt[0][1]) + 2.0f * z * (dMt[2][0] + dMt[0][2]) + 2.0f * r * (dMt[1][2] - dMt[2][1]) - 4.0f * x * (dMt[2][2] + dMt[1][1]);
        o.rot[2] = 2.0f * x * (dMt[1][0] + dMt[0][1]) + 2.0f * r * (dMt[2][0] - dMt[0][2]) + 2.0f * z * (dMt[1][2] + dMt[2][1]) - 4.0f * y * (dMt[2][2] + dMt[0][0]);
        o.rot[3] = 2.0f * r * (dMt[0][1] - dMt[1][0]) + 2.0f * x * (dMt[2][0] + dMt[0][2]) + 2.0f * y * (dMt[1][2] + dMt[2][1]) - 4.0f * z * (dMt[1][1] + dMt[0][0]);
    }
}

// Sum the nine pixel-stage partials of splat i over its `tiles` slots, ascending slot order.  With row marks (epoch != 0: cameras
// with long tile lists, uses_row_marks; the seam always) a slot holds a row only if its mark
// equals the launch's epoch (k_render.hip: the backward writes and marks rows for evaluated entries only); the others are
// implicit zero rows.  The marks are fetched one trip ahead of the rows they guard.  The first trip cannot know its marks
// before it asks for its rows: it requests marks and rows together (slots behind the splat's last one are redirected to
// `zero` by selecting the ADDRESS) and selects afterwards — the only absent rows ever read, at most GS_GATHER_ROWS per splat;
// cfg3's splats touch ~7 tiles, so most of their rows are in this trip and none of them waits for a mark.
__device__ inline void gather_rows(const float* __restrict__ Gv, const uint8_t* __restrict__ marks, uint32_t epoch, const float* __restrict__ zero,
                                   uint32_t first, uint32_t tiles, float sum[9]) {
#pragma unroll
    for (int q = 0; q < 9; q++) sum[q] = 0.0f;
    const Row3* row = reinterpret_cast<const Row3*>(Gv + (size_t)first * G_STRIDE);
    const Row3* zrow = reinterpret_cast<const Row3*>(zero);
    const uint8_t* mk = marks + first;
    // GS_GATHER_ROWS rows in flight per trip (the loop is latency-bound: a splat touches ~7 tiles); the adds keep slot order
#ifndef GS_GATHER_ROWS
#define GS_GATHER_ROWS 4  // tuning hook (tools/build_variant.sh)
#endif
    constexpr int GR = GS_GATHER_ROWS;
    auto add_rows = [&](const Row3 (&r)[3 * GR]) {   // adding the zeros of an absent row changes nothing: the sums start at +0 and never become -0
#pragma unroll
        for (int j = 0; j < GR; j++) {
            sum[0] += r[3 * j].a; sum[1] += r[3 * j].b; sum[2] += r[3 * j].c;
            sum[3] += r[3 * j + 1].a; sum[4] += r[3 * j + 1].b; sum[5] += r[3 * j + 1].c;
            sum[6] += r[3 * j + 2].a; sum[7] += r[3 * j + 2].b; sum[8] += r[3 * j + 2].c;
        }
    };
    Row3 r[3 * GR];
    if (epoch == 0) {   // no marks this launch (short lists: nearly every row exists): every slot holds a row
        uint32_t k = 0;
        for (; k + GR <= tiles; k += GR, row += 3 * GR) {
#pragma unroll
            for (int j = 0; j < 3 * GR; j++) r[j] = row[j];
            add_rows(r);
        }
        for (; k < tiles; k++, row += 3) {
            const Row3 r0 = row[0], r1 = row[1], r2 = row[2];
            sum[0] += r0.a; sum[1] += r0.b; sum[2] += r0.c; sum[3] += r1.a; sum[4] += r1.b; sum[5] += r1.c;
            sum[6] += r2.a; sum[7] += r2.b; sum[8] += r2.c;
        }
        return;
    }
    uint8_t m[GR], m_next[GR];
    // trip 0: marks, rows and the next trip's marks are all requested at once
#pragma unroll
    for (int j = 0; j < GR; j++) {
        const bool in = (uint32_t)j < tiles;
        m[j] = in ? mk[j] : (uint8_t)0;
        const Row3* src = in ? row + 3 * j : zrow;
        r[3 * j] = src[0]; r[3 * j + 1] = src[1]; r[3 * j + 2] = src[2];
        m_next[j] = (uint32_t)(GR + j) < tiles ? mk[GR + j] : (uint8_t)0;
    }
#pragma unroll
    for (int j = 0; j < GR; j++) {
        const bool have = (uint32_t)m[j] == epoch;   // (a slot behind the splat's last one reads as mark 0: never an epoch)
#pragma unroll
        for (int q = 0; q < 3; q++) { r[3 * j + q].a = have ? r[3 * j + q].a : 0.0f; r[3 * j + q].b = have ? r[3 * j + q].b : 0.0f; r[3 * j + q].c = have ? r[3 * j + q].c : 0.0f; }
    }
    add_rows(r);
    row += 3 * GR;
    for (uint32_t k = GR; k < tiles; k += GR, row += 3 * GR) {
#pragma unroll
        for (int j = 0; j < GR; j++) { m[j] = m_next[j]; m_next[j] = k + GR + j < tiles ? mk[k + GR + j] : (uint8_t)0; }
        // from the second trip on the marks are known before the rows are asked for: an absent row is not requested at all (a
        // dense scene — 84 % of the rows absent — is bound by the number of requests, which reading `zero` instead would keep)
#pragma unroll
        for (int j = 0; j < GR; j++) {
            const Row3 z{ 0.0f, 0.0f, 0.0f };
            r[3 * j] = z; r[3 * j + 1] = z; r[3 * j + 2] = z;
            if ((uint32_t)m[j] == epoch) { r[3 * j] = row[3 * j]; r[3 * j + 1] = row[3 * j + 1]; r[3 * j + 2] = row[3 * j + 2]; }
        }
        add_rows(r);
    }
}

// SH basis exactly as the backward writes it: dL_dsh[k][c] = basis[k] * dL_dRGB[c].
template <int D> __device__ inline void sh_basis(float X, float Y, float Z, float* b) {
    b[0] = C0;
    if constexpr (D > 0) { b[1] = -C1 * Y; b[2] = C1 * Z; b[3] = -C1 * X; }
    if constexpr (D > 1) {
        const float xx = X * X, yy = Y * Y, zz = Z * Z, xy = X * Y, yz = Y * Z, xz = X * Z;
        b[4] = C2_0 * xy; b[5] = C2_1 * yz; b[6] = C2_2 * (2.0f * zz - xx - yy); b[7] = C2_3 * xz; b[8] = C2_4 * (xx - yy);
        if constexpr (D > 2) {
            b[9] = C3_0 * Y * (3.0f * xx - yy); b[10] = C3_1 * xy * Z; b[11] = C3_2 * Y * (4.0f * zz - xx - yy);
            b[12] = C3_3 * Z * (2.0f * zz - 3.0f * xx - 3.0f * yy); b[13] = C3_4 * X * (4.0f * zz - xx - yy);
            b[14] = C3_5 * Z * (xx - yy); b[15] = C3_6 * X * (xx - 3.0f * yy);
        }
    }
}

// accumulateGradients (src/Trainer.cu:51-76) for one splat: var += |g_loc| / S, avg += g / S over the records of its passes
// in the reference's order; the SH gradient is rebuilt as basis(view direction) x dL_dRGB.
template <int D> struct GradAcc {
    static constexpr int NC = (D + 1) * (D + 1);
    float var = 0.0f, aLoc[3] = { 0, 0, 0 }, aScale[3] = { 0, 0, 0 }, aRot[4] = { 0, 0, 0, 0 }, aOpac = 0.0f, aSh[NC][3];
    __device__ GradAcc() {
#pragma unroll
        for (int k = 0; k < NC; k++) aSh[k][0] = aSh[k][1] = aSh[k][2] = 0.0f;
    }
    // one record: mean(3) scale(3) rot(4) opacity(1) pad | dL_dRGB(3) pad — the colour gradient fills a 16-byte quarter of its own, so the
    // reduction's SH parts fetch that quarter alone; (mx, my, mz) the splat, cp the pass's camera position
    __device__ void add(const float4& r0, const float4& r1, const float4& r2, const float4& r3, float samples, float mx, float my, float mz,
                        const float* cp) {
        add_geometry(r0, r1, r2, samples);
        const float dRGB[3] = { r3.x, r3.y, r3.z };
        add_sh(dRGB, samples, mx, my, mz, cp);
    }
    // everything of accumulateGradients but the SH planes (the compact data-parallel exchange sums these per rank and all-reduces them)
    __device__ void add_geometry(const float4& r0, const float4& r1, const float4& r2, float samples) {
        const float gm[3] = { r0.x, r0.y, r0.z }, gs3[3] = { r0.w, r1.x, r1.y }, gr[4] = { r1.z, r1.w, r2.x, r2.y };
        var += sqrtf((gm[0] * gm[0]) + (gm[1] * gm[1]) + (gm[2] * gm[2])) / samples;
#pragma unroll
        for (int c = 0; c < 3; c++) { aLoc[c] += gm[c] / samples; aScale[c] += gs3[c] / samples; }
        aOpac += r2.z / samples;
#pragma unroll
        for (int c = 0; c < 4; c++) aRot[c] += gr[c] / samples;
    }
    // the SH planes' share of one record: avg_sh += (basis(direction from the record's camera) x dL_dRGB) / S
    __device__ void add_sh(const float (&dRGB)[3], float samples, float mx, float my, float mz, const float* cp) {
        // A pass in which the splat is culled (or its colour gradient is exactly zero) adds +0 to every SH sum: skip it,
        // as upstream's radii > 0 guard does — the basis of a splat AT the camera position (len = 0) or with a
        // non-finite mean is NaN, and NaN * 0 would poison the SH planes for good.
        if (dRGB[0] != 0.0f || dRGB[1] != 0.0f || dRGB[2] != 0.0f) {
            const float ox = mx - cp[0], oy = my - cp[1], oz = mz - cp[2];
            const float len = sqrtf(ox * ox + oy * oy + oz * oz);
            float basis[NC];
            sh_basis<D>(ox / len, oy / len, oz / len, basis);
#pragma unroll
            for (int k = 0; k < NC; k++)
#pragma unroll
                for (int c = 0; c < 3; c++) aSh[k][c] += (basis[k] * dRGB[c]) / samples;
        }
    }
    // the twelve planes that are not SH, in the exchange's order: loc 3 | scale 3 | opacity | rot 4 | var (Exchange::geo)
    __device__ void store_geometry(float* __restrict__ geo, size_t st, int i, bool zero_var) const {
#pragma unroll
        for (int c = 0; c < 3; c++) { geo[c * st + i] = aLoc[c]; geo[(3 + c) * st + i] = aScale[c]; }
        geo[6 * st + i] = aOpac;
#pragma unroll
        for (int c = 0; c < 4; c++) geo[(7 + c) * st + i] = aRot[c];
        geo[11 * st + i] = zero_var ? 0.0f : var;
    }
    // every gradient plane of splat i, written exactly once per step
    __device__ void store(float* __restrict__ grad, const Planes& pl, size_t st, int i, int M, bool zero_var) const {
#pragma unroll
        for (int c = 0; c < 3; c++) { grad[pl.loc(c) * st + i] = aLoc[c]; grad[pl.scale(c) * st + i] = aScale[c]; }
#pragma unroll
        for (int k = 0; k < NC; k++)
#pragma unroll
            for (int c = 0; c < 3; c++) grad[pl.sh(k, c) * st + i] = aSh[k][c];
        for (int k = NC; k < M; k++)
            for (int c = 0; c < 3; c++) grad[pl.sh(k, c) * st + i] = 0.0f;
        grad[pl.opac() * st + i] = aOpac;
#pragma unroll
        for (int c = 0; c < 4; c++) grad[pl.rot(c) * st + i] = aRot[c];
        grad[pl.var() * st + i] = zero_var ? 0.0f : var;
    }
};

// Trainer stage 1: one thread per (pass, splat) — fully parallel.  Sums the splat's gradient rows of that pass and
// runs the per-splat chain; the result is ONE 64-byte record per (pass, splat):
//   mean(3) scale(3) rot(4) opacity(1) pad(1) | dL_dRGB(3, clamp-masked) pad(1).   Culled splats write zeros.
// (Fusing the two passes of a camera into one thread was measured slower: 175 VGPRs, 2 waves/SIMD.)
// Work items are the backward's {group, pass a, pass b}.  With fused pairs (render_bwd<2>) there is one
// gradient set per item, in pass a's slice: blockIdx.y then enumerates items and the record is pass a's.
// SINGLE: the step has exactly one record per splat (one work item: the per-GPU load of an 8-GPU run) — the record never leaves
// the registers: accumulateGradients is applied here and the gradient planes are written, no second launch.
// Where a record's dL_dRGB goes in the compact exchange's gather buffer (Exchange::rgb): chunk `rank`, slot, three planes of Pa.
__device__ inline float* exchange_chunk(const Exchange& x, int rank, size_t Pa) { return x.rgb + (size_t)rank * ((size_t)x.hdr + (size_t)x.slots * 3 * Pa); }
__device__ inline float* exchange_rgb(const Exchange& x, int rank, int slot, size_t Pa) { return exchange_chunk(x, rank, Pa) + x.hdr + (size_t)slot * 3 * Pa; }
// The header of this rank's chunk: the position of each of its cameras (local camera = geometry group), from the view block the step
// rendered with.  One workgroup writes it; it leaves the rank with the records in the same all-gather.
__device__ inline void exchange_write_header(const Dims& d, const Scratch& s, const Exchange& x) {
    if (blockIdx.x != 0 || blockIdx.y != 0) return;
    float* h = exchange_chunk(x, x.rank, (size_t)d.Pa);
    for (int t = threadIdx.x; t < x.hdr; t += WG) { const int g = t / 3; h[t] = g < d.VG ? s.gviews[g].campos[t - 3 * g] : 0.0f; }
}

// SINGLE + x.geo != null (compact exchange): the geometry sums go to the exchange's twelve planes and the record's dL_dRGB to
// its slot 0 of this rank's chunk of the gather buffer; no SH plane is touched (k_sh_rebuild writes them after the exchange).
#ifndef GS_SBV_WAVES
#define GS_SBV_WAVES 0  // occupancy target of k_splat_bwd_view, waves per SIMD (tuning hook, tools/build_variant.sh; 0: the compiler's choice)
#endif
#if GS_SBV_WAVES > 0
#define GS_SBV_ATTR __attribute__((amdgpu_waves_per_eu(GS_SBV_WAVES, GS_SBV_WAVES)))
#else
#define GS_SBV_ATTR
#endif
template <int D, bool SINGLE>
__global__ __launch_bounds__(WG) GS_SBV_ATTR void k_splat_bwd_view(Dims d, const float* __restrict__ params, Scratch s, float4* __restrict__ rec_out,
                                                       const int* __restrict__ items, int n_pairs, int fused, float samples,
                                                       float* __restrict__ grad, Exchange x) {
    const int i = blockIdx.x * WG + threadIdx.x;
    if constexpr (SINGLE) { if (x.geo) exchange_write_header(d, s, x); }
    if (i >= d.P) return;
    // blockIdx.y enumerates passes: the two passes of every pair item first, then the single items
    const int y = blockIdx.y;
    const int which = (!fused && y < 2 * n_pairs) ? (y & 1) : 0;
    const int* item = items + 3 * (fused ? y : (y < 2 * n_pairs ? (y >> 1) : (y - n_pairs)));
    const int g = item[0];          // geometry group: records and slots live there
    const int v = item[1 + which];  // pass: gradient rows and the output record
    const Planes pl{ d.M };
    const size_t st = (size_t)d.Pa;
    float4* out = rec_out + ((size_t)v * st + i) * 4;
    // culled splats touch no tile (k_preprocess); everything this kernel needs of the projection is in tiles_touched,
    // point_offsets and the 48-byte sh_jac record — the 64-byte GeomRec is not read here
    const uint32_t tiles = s.tiles_touched[(size_t)g * st + i];
    if ((s.flags[g * 4 + 0] & 3u) || tiles == 0) {  // culled (or a group the host will replay): the reference's nine buffers stay zero
        const float4 z = make_float4(0, 0, 0, 0);
        if constexpr (SINGLE) {
            GradAcc<D> acc;
            if (x.geo) {
                acc.store_geometry(x.geo, st, i, fused != 0);
                float* rgb = exchange_rgb(x, x.rank, 0, st);
                rgb[i] = 0.0f; rgb[st + i] = 0.0f; rgb[2 * st + i] = 0.0f;
            } else acc.store(grad, pl, st, i, d.M, fused != 0);
        } else { out[0] = z; out[1] = z; out[2] = z; out[3] = z; }
        return;
    }
    float mean[3], sc[3], q[4];
#pragma unroll
    for (int c = 0; c < 3; c++) { mean[c] = params[pl.loc(c) * st + i]; sc[c] = d.mod * params[pl.scale(c) * st + i]; }
#pragma unroll
    for (int c = 0; c < 4; c++) q[c] = params[pl.rot(c) * st + i];
    const uint32_t first = s.point_offsets[(size_t)g * st + i] - tiles;
    float sum[9];
    gather_rows(s.G + (size_t)v * d.Rcap * G_STRIDE, s.row_epoch + (size_t)v * d.Rcap, uses_row_marks(d, s, g) ? (uint32_t)d.epoch : 0u, s.zero_row, first, tiles, sum);
    SplatOut<D> o;
    float dRGB[3];
    // d colour / d direction was evaluated by the projection (same camera for every pass of the group): 9 floats
    // instead of 3*M SH coefficients per (pass, splat)
    const float4* jrec = reinterpret_cast<const float4*>(s.sh_jac + ((size_t)g * st + i) * 12);
    const float4 j0 = jrec[0], j1 = jrec[1], j2 = jrec[2];
    const float jv[9] = { j0.x, j0.y, j0.z, j0.w, j1.x, j1.y, j1.z, j1.w, j2.x };
    auto jac_at = [&](int ch, float, float, float, float& dx_, float& dy_, float& dz_) { dx_ = jv[3 * ch]; dy_ = jv[3 * ch + 1]; dz_ = jv[3 * ch + 2]; };
    splat_backward_core<D, false>(s.gviews[g], d.W, d.H, mean, sc, q, jac_at, __float_as_uint(j2.y), sum, sum[3], sum[4], sum[5], sum[6], sum[7], o, dRGB);
    const float4 r0 = make_float4(o.mean[0], o.mean[1], o.mean[2], o.scale[0]), r1 = make_float4(o.scale[1], o.scale[2], o.rot[0], o.rot[1]),
                 r2 = make_float4(o.rot[2], o.rot[3], sum[8], 0.0f), r3 = make_float4(dRGB[0], dRGB[1], dRGB[2], 0.0f);
    if constexpr (SINGLE) {
        GradAcc<D> acc;
        if (x.geo) {
            acc.add_geometry(r0, r1, r2, samples);
            acc.store_geometry(x.geo, st, i, fused != 0);
            float* rgb = exchange_rgb(x, x.rank, 0, st);
            rgb[i] = dRGB[0]; rgb[st + i] = dRGB[1]; rgb[2 * st + i] = dRGB[2];
        } else {
            acc.add(r0, r1, r2, r3, samples, mean[0], mean[1], mean[2], s.views[v].campos);
            acc.store(grad, pl, st, i, d.M, fused != 0);
        }
    } else {
        out[0] = r0; out[1] = r1; out[2] = r2; out[3] = r3;
    }
}

// Trainer stage 2: one thread per splat walks the views in the reference's order and applies
// accumulateGradients (src/Trainer.cu:51-76: var += |g_loc| / S, avg += g / S); the SH gradient is rebuilt as
// basis(view direction) x dL_dRGB.  Every gradient plane is written exactly once per step.
// n_fused_items > 0: the records are one per work item (the pair's summed gradient, at the item's first pass) and
// `var`, which needs every pass's own location gradient, is written as zero (see render_bwd in k_render.hip).
// The reduction is split into PARTS so that a step's 100 000 splats are more than 1.5 waves per SIMD: blockIdx.y = 0 owns the
// twelve planes that are not SH (it reads 48 bytes of every record), blockIdx.y = 1 .. owns a range of SH coefficients each (the
// record's dL_dRGB quarter, 16 bytes, + the view direction).  Every plane's sum runs over the records in the reference's order
// exactly as one thread per splat ran it: the same bits.  (One thread per splat: 35 us at cfg3 for 75 MB — latency, not bytes.)
// UPD (a step with no collective between the reduction and the update): the thread applies the update of ITS planes itself, with
// update_value — k_update's operations on the value k_update would have read back from the gradient plane — parameters and moments
// loaded a batch of planes at a time through restrict-qualified pointers (the compiler must not order every load behind the previous
// plane's stores).  The SH planes beyond the model's degree get their zero gradient (Adam moves on a zero gradient too).
// All parameter reads of the UPD kernel go through fu.params (it writes them).
template <int N>
__device__ inline void update_planes(const FusedUpdate& fu, const Planes& pl, const int (&plane)[N], const float (&g)[N], size_t st, int i) {
    float* __restrict__ P = fu.params;
    float* __restrict__ Am = fu.am;
    float* __restrict__ Av = fu.av;
    __half* __restrict__ H = reinterpret_cast<__half*>(fu.sh16);
    const bool adam = fu.u.rule == GS_UPDATE_ADAM;
    float x[N], m[N], v[N];
#pragma unroll
    for (int k = 0; k < N; k++) {
        const size_t idx = (size_t)plane[k] * st + i;
        x[k] = P[idx];
        m[k] = adam ? Am[idx] : 0.0f;
        v[k] = adam ? Av[idx] : 0.0f;
    }
#pragma unroll
    for (int k = 0; k < N; k++) {
        float lr; int kind;
        update_plane_rule(fu.u, pl, plane[k], lr, kind);
        x[k] = update_value(fu.u, lr, kind, x[k], g[k], m[k], v[k]);
    }
#pragma unroll
    for (int k = 0; k < N; k++) {
        const size_t idx = (size_t)plane[k] * st + i;
        P[idx] = x[k];
        if (adam) { Am[idx] = m[k]; Av[idx] = v[k]; }
        if (H && plane[k] >= 3 && plane[k] < pl.scale(0)) H[(size_t)(plane[k] - 3) * st + i] = __float2half_rn(x[k]);
    }
}

// SH coefficients [K0, K1) of splat i: avg_sh[k][c] += (basis[k] * dL_dRGB[c]) / S over the records, GradAcc::add_sh's operations
template <int D, int K0, int K1, bool UPD>
__device__ inline void reduce_sh_part(const Dims& d, const Scratch& s, float samples, const float4* __restrict__ rec_in, float* __restrict__ grad,
                                      const int* __restrict__ items, int n_fused_items, const FusedUpdate& fu, bool apply, bool last_part,
                                      float mx, float my, float mz, int i) {
    constexpr int NC = (D + 1) * (D + 1), NK = K1 - K0;
    const Planes pl{ d.M };
    const size_t st = (size_t)d.Pa;
    float a[NK][3];
#pragma unroll
    for (int k = 0; k < NK; k++) a[k][0] = a[k][1] = a[k][2] = 0.0f;
    const int n_rec = n_fused_items > 0 ? n_fused_items : d.V;
    for (int r = 0; r < n_rec; r++) {
        const int v = n_fused_items > 0 ? items[3 * r + 1] : r;
        const float4 q = rec_in[((size_t)v * st + i) * 4 + 3];
        if (q.x != 0.0f || q.y != 0.0f || q.z != 0.0f) {       // (as GradAcc::add_sh: a culled record adds +0, and its basis may be NaN)
            const float* cp = s.views[v].campos;
            const float ox = mx - cp[0], oy = my - cp[1], oz = mz - cp[2];
            const float len = sqrtf(ox * ox + oy * oy + oz * oz);
            float basis[NC];
            sh_basis<D>(ox / len, oy / len, oz / len, basis);
            const float dRGB[3] = { q.x, q.y, q.z };
#pragma unroll
            for (int k = 0; k < NK; k++)
#pragma unroll
                for (int c = 0; c < 3; c++) a[k][c] += (basis[K0 + k] * dRGB[c]) / samples;
        }
    }
    int plane[3 * NK];
    float g[3 * NK];
#pragma unroll
    for (int k = 0; k < NK; k++)
#pragma unroll
        for (int c = 0; c < 3; c++) { plane[3 * k + c] = pl.sh(K0 + k, c); g[3 * k + c] = a[k][c]; grad[pl.sh(K0 + k, c) * st + i] = a[k][c]; }
    if (last_part)
        for (int k = NC; k < d.M; k++)
            for (int c = 0; c < 3; c++) grad[pl.sh(k, c) * st + i] = 0.0f;
    if constexpr (UPD) {
        if (!apply) return;
        update_planes<3 * NK>(fu, pl, plane, g, st, i);
        if (last_part)
            for (int k = NC; k < d.M; k++) {
                const int pz[3] = { pl.sh(k, 0), pl.sh(k, 1), pl.sh(k, 2) };
                const float gz[3] = { 0.0f, 0.0f, 0.0f };
                update_planes<3>(fu, pl, pz, gz, st, i);
            }
    }
}

template <int D> constexpr int reduce_parts() { return D == 3 ? 4 : (D == 2 ? 3 : 2); }    // geometry + 1 / 2 / 3 SH coefficient ranges

template <int D, bool UPD>
__global__ __launch_bounds__(WG) void k_splat_bwd_reduce(Dims d, const float* __restrict__ params_ro, Scratch s, float samples,
                                                         const float4* __restrict__ rec_in, float* __restrict__ grad,
                                                         const int* __restrict__ items, int n_fused_items, FusedUpdate fu) {
    const int i = blockIdx.x * WG + threadIdx.x;
    if (i >= d.P) return;
    const Planes pl{ d.M };
    const size_t st = (size_t)d.Pa;
    const int part = blockIdx.y;
    bool apply = false;
    if constexpr (UPD) {
        // an attempt whose binning arena overflowed is replayed by the host with a larger one: it must not move the model
        uint32_t overflow = 0;
        for (int g = 0; g < d.VG; g++) overflow |= s.flags[g * 4 + 0] & 3u;     // (bit 1: a depth cut the forward found wrong)
        apply = overflow == 0;
    }
    if (part == 0) {
        GradAcc<0> acc;      // (its one SH coefficient stays unused here)
        const int n_rec = n_fused_items > 0 ? n_fused_items : d.V;
        for (int k = 0; k < n_rec; k++) {
            const int v = n_fused_items > 0 ? items[3 * k + 1] : k;
            const float4* r = rec_in + ((size_t)v * st + i) * 4;
            acc.add_geometry(r[0], r[1], r[2], samples);
        }
        const bool zero_var = n_fused_items > 0;
        int plane[11];
        float g[11];
#pragma unroll
        for (int c = 0; c < 3; c++) { plane[c] = pl.loc(c); g[c] = acc.aLoc[c]; plane[3 + c] = pl.scale(c); g[3 + c] = acc.aScale[c]; }
        plane[6] = pl.opac(); g[6] = acc.aOpac;
#pragma unroll
        for (int c = 0; c < 4; c++) { plane[7 + c] = pl.rot(c); g[7 + c] = acc.aRot[c]; }
#pragma unroll
        for (int k = 0; k < 11; k++) grad[(size_t)plane[k] * st + i] = g[k];
        grad[pl.var() * st + i] = zero_var ? 0.0f : acc.var;
        if constexpr (UPD) { if (apply) update_planes<11>(fu, pl, plane, g, st, i); }
        return;
    }
    // The SH parts read the splat's position: in the fused form the geometry part of the SAME splat may have moved it already (another
    // workgroup), so the position the gradients were formed at is taken from where nobody writes — the projection's record of it.
    float mx, my, mz;
    if constexpr (UPD) { mx = s.mean_copy[i]; my = s.mean_copy[st + i]; mz = s.mean_copy[2 * st + i]; }
    else { mx = params_ro[pl.loc(0) * st + i]; my = params_ro[pl.loc(1) * st + i]; mz = params_ro[pl.loc(2) * st + i]; }
    if constexpr (D == 0) reduce_sh_part<0, 0, 1, UPD>(d, s, samples, rec_in, grad, items, n_fused_items, fu, apply, true, mx, my, mz, i);
    else if constexpr (D == 1) reduce_sh_part<1, 0, 4, UPD>(d, s, samples, rec_in, grad, items, n_fused_items, fu, apply, true, mx, my, mz, i);
    else if constexpr (D == 2) {
        if (part == 1) reduce_sh_part<2, 0, 5, UPD>(d, s, samples, rec_in, grad, items, n_fused_items, fu, apply, false, mx, my, mz, i);
        else reduce_sh_part<2, 5, 9, UPD>(d, s, samples, rec_in, grad, items, n_fused_items, fu, apply, true, mx, my, mz, i);
    } else {
        if (part == 1) reduce_sh_part<3, 0, 6, UPD>(d, s, samples, rec_in, grad, items, n_fused_items, fu, apply, false, mx, my, mz, i);
        else if (part == 2) reduce_sh_part<3, 6, 11, UPD>(d, s, samples, rec_in, grad, items, n_fused_items, fu, apply, false, mx, my, mz, i);
        else reduce_sh_part<3, 11, 16, UPD>(d, s, samples, rec_in, grad, items, n_fused_items, fu, apply, true, mx, my, mz, i);
    }
}

// The compact data-parallel exchange (capi.hip, gs_trainer_set_compact_exchange), rank side: the rank's records are summed into
// the twelve geometry planes it contributes to the all-reduce, and every record's dL_dRGB goes to its slot of the rank's chunk
// of the all-gather buffer.  Records: one per local camera g in the fused form (slot g), one per local pass in the per-pass
// form (first pass of local camera g — its white one: slot g; the other: slot x.slots / 2 + g).  Slots the rank does not fill
// are zeroed.
__global__ __launch_bounds__(WG) void k_exchange_pack(Dims d, Scratch s, float samples, const float4* __restrict__ rec_in, Exchange x,
                                                      const int* __restrict__ items, int n_fused_items) {
    const int i = blockIdx.x * WG + threadIdx.x;
    exchange_write_header(d, s, x);
    if (i >= d.P) return;
    const size_t st = (size_t)d.Pa;
    GradAcc<0> acc;
    const int n_rec = n_fused_items > 0 ? n_fused_items : d.V;
    for (int k = 0; k < n_rec; k++) {
        const int v = n_fused_items > 0 ? items[3 * k + 1] : k;
        const float4* r = rec_in + ((size_t)v * st + i) * 4;
        const float4 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
        acc.add_geometry(r0, r1, r2, samples);
        // local camera = the pass's geometry group (groups are numbered in the order of their first passes); a group's first
        // pass is its white one in the reference's order (src/Trainer.cu:311-314)
        const int g = n_fused_items > 0 ? items[3 * k] : s.view_group[v];
        const int slot = (n_fused_items > 0 || s.group_views[s.group_first[g]] == v) ? g : x.slots / 2 + g;
        float* rgb = exchange_rgb(x, x.rank, slot, st);
        rgb[i] = r3.x; rgb[st + i] = r3.y; rgb[2 * st + i] = r3.z;
    }
    // a rank with fewer cameras than the others (cameras % ranks != 0) sends zeros in the slots it has no record for
    const int per_half = n_fused_items > 0 ? x.slots : x.slots / 2;
    for (int h = 0; h < (n_fused_items > 0 ? 1 : 2); h++)
        for (int k = d.VG; k < per_half; k++) {
            float* rgb = exchange_rgb(x, x.rank, h * per_half + k, st);
            rgb[i] = 0.0f; rgb[st + i] = 0.0f; rgb[2 * st + i] = 0.0f;
        }
    acc.store_geometry(x.geo, st, i, n_fused_items > 0);
}

// ... and after the collectives, on every rank alike: the averaged-gradient planes of the whole iteration.  The twelve geometry
// planes are the all-reduced sums; the SH planes are rebuilt from the gathered dL_dRGB of EVERY record of the iteration in the
// order a single GPU walks them (fused form: cameras 0 .. C-1; per-pass form: their white passes, then their black ones —
// src/Trainer.cu:311-314), each with its own camera's view direction: avg_sh += (basis x dL_dRGB) / S, the very operations of
// GradAcc::add — the SH gradients are the single-GPU step's bit for bit, whatever order the collective summed in.
// Camera c of the iteration lives on rank c % world as that rank's local camera c / world.
// parts: bit 0 = the SH planes (needs the gathered records only), bit 1 = the twelve other planes (needs the all-reduce only):
// with the two collectives side by side the SH planes are rebuilt while the all-reduce is still under way.
// In PARTS like k_splat_bwd_reduce (blockIdx.y + first_part: 0 = the twelve other planes, 1 .. = SH coefficient ranges), and with UPD the
// thread applies the update of its planes itself (FusedUpdate; the host only asks for it once the step's attempt stands, so there is no
// flag to look at here): the data-parallel step then has no update launch either, and the gradient planes are not read back.
template <int D, int K0, int K1, bool UPD>
__device__ inline void rebuild_sh_part(const Dims& d, const Exchange& x, int n_cameras, int per_pass, float samples, float* __restrict__ grad,
                                       const FusedUpdate& fu, bool last_part, float mx, float my, float mz, int i) {
    constexpr int NC = (D + 1) * (D + 1), NK = K1 - K0;
    const Planes pl{ d.M };
    const size_t st = (size_t)d.Pa;
    float a[NK][3];
#pragma unroll
    for (int k = 0; k < NK; k++) a[k][0] = a[k][1] = a[k][2] = 0.0f;
    const int n_rec = per_pass ? 2 * n_cameras : n_cameras;
    for (int r = 0; r < n_rec; r++) {
        const int c = r < n_cameras ? r : r - n_cameras;
        const int slot = (r < n_cameras ? 0 : x.slots / 2) + c / x.world;
        const float* rgb = exchange_rgb(x, c % x.world, slot, st);
        const float dRGB[3] = { rgb[i], rgb[st + i], rgb[2 * st + i] };
        if (dRGB[0] != 0.0f || dRGB[1] != 0.0f || dRGB[2] != 0.0f) {        // (GradAcc::add_sh)
            const float* cp = exchange_chunk(x, c % x.world, st) + 3 * (c / x.world);   // the position that came with the record
            const float ox = mx - cp[0], oy = my - cp[1], oz = mz - cp[2];
            const float len = sqrtf(ox * ox + oy * oy + oz * oz);
            float basis[NC];
            sh_basis<D>(ox / len, oy / len, oz / len, basis);
#pragma unroll
            for (int k = 0; k < NK; k++)
#pragma unroll
                for (int ch = 0; ch < 3; ch++) a[k][ch] += (basis[K0 + k] * dRGB[ch]) / samples;
        }
    }
    int plane[3 * NK];
    float g[3 * NK];
#pragma unroll
    for (int k = 0; k < NK; k++)
#pragma unroll
        for (int ch = 0; ch < 3; ch++) { plane[3 * k + ch] = pl.sh(K0 + k, ch); g[3 * k + ch] = a[k][ch]; grad[pl.sh(K0 + k, ch) * st + i] = a[k][ch]; }
    if (last_part)
        for (int k = NC; k < d.M; k++)
            for (int ch = 0; ch < 3; ch++) grad[pl.sh(k, ch) * st + i] = 0.0f;
    if constexpr (UPD) {
        update_planes<3 * NK>(fu, pl, plane, g, st, i);
        if (last_part)
            for (int k = NC; k < d.M; k++) {
                const int pz[3] = { pl.sh(k, 0), pl.sh(k, 1), pl.sh(k, 2) };
                const float gz[3] = { 0.0f, 0.0f, 0.0f };
                update_planes<3>(fu, pl, pz, gz, st, i);
            }
    }
}

template <int D, bool UPD>
__global__ __launch_bounds__(WG) void k_sh_rebuild(Dims d, const float* __restrict__ params_ro, const float* __restrict__ mean_copy, Exchange x,
                                                   int n_cameras, int per_pass, float samples, float* __restrict__ grad, int first_part, FusedUpdate fu) {
    const int i = blockIdx.x * WG + threadIdx.x;
    if (i >= d.P) return;
    const Planes pl{ d.M };
    const size_t st = (size_t)d.Pa;
    const int part = first_part + blockIdx.y;
    if (part == 0) {
        int plane[11];
        float g[11];
#pragma unroll
        for (int c = 0; c < 3; c++) { plane[c] = pl.loc(c); g[c] = x.geo[c * st + i]; plane[3 + c] = pl.scale(c); g[3 + c] = x.geo[(3 + c) * st + i]; }
        plane[6] = pl.opac(); g[6] = x.geo[6 * st + i];
#pragma unroll
        for (int c = 0; c < 4; c++) { plane[7 + c] = pl.rot(c); g[7 + c] = x.geo[(7 + c) * st + i]; }
#pragma unroll
        for (int k = 0; k < 11; k++) grad[(size_t)plane[k] * st + i] = g[k];
        grad[pl.var() * st + i] = x.geo[11 * st + i];
        if constexpr (UPD) update_planes<11>(fu, pl, plane, g, st, i);
        return;
    }
    // (UPD: the geometry part of the same splat may already have moved the position plane: the SH parts take it from the projection's copy)
    float mx, my, mz;
    if constexpr (UPD) { mx = mean_copy[i]; my = mean_copy[st + i]; mz = mean_copy[2 * st + i]; }
    else { mx = params_ro[pl.loc(0) * st + i]; my = params_ro[pl.loc(1) * st + i]; mz = params_ro[pl.loc(2) * st + i]; }
    if constexpr (D == 0) rebuild_sh_part<0, 0, 1, UPD>(d, x, n_cameras, per_pass, samples, grad, fu, true, mx, my, mz, i);
    else if constexpr (D == 1) rebuild_sh_part<1, 0, 4, UPD>(d, x, n_cameras, per_pass, samples, grad, fu, true, mx, my, mz, i);
    else if constexpr (D == 2) {
        if (part == 1) rebuild_sh_part<2, 0, 5, UPD>(d, x, n_cameras, per_pass, samples, grad, fu, false, mx, my, mz, i);
        else rebuild_sh_part<2, 5, 9, UPD>(d, x, n_cameras, per_pass, samples, grad, fu, true, mx, my, mz, i);
    } else {
        if (part == 1) rebuild_sh_part<3, 0, 6, UPD>(d, x, n_cameras, per_pass, samples, grad, fu, false, mx, my, mz, i);
        else if (part == 2) rebuild_sh_part<3, 6, 11, UPD>(d, x, n_cameras, per_pass, samples, grad, fu, false, mx, my, mz, i);
        else rebuild_sh_part<3, 11, 16, UPD>(d, x, n_cameras, per_pass, samples, grad, fu, true, mx, my, mz, i);
    }
}

// x != null: the compact exchange's rank side (the rank holds whole cameras: n1 == 0); grad is not written
template <int D>
static void launch_splat_avg_d(const Dims& d, const float* params, const Scratch& s, float samples, float* grad, const int* items, int n2,
                               int n1, bool fuse, const Exchange* x, hipStream_t stream, const FusedUpdate* fu, bool* update_applied) {
    const int bx = (d.P + WG - 1) / WG;
    float4* rec = reinterpret_cast<float4*>(s.splat_grads);
    fuse = fuse && n2 > 0;
    const int ny = fuse ? n2 + n1 : 2 * n2 + n1;
    const Exchange none{};
    if (ny == 1) {  // one record per splat: chain and accumulateGradients in one launch
        hipLaunchKernelGGL((k_splat_bwd_view<D, true>), dim3(bx, 1), dim3(WG), 0, stream, d, params, s, rec, items, n2, fuse ? 1 : 0, samples, grad, x ? *x : none);
        return;
    }
    if (ny > 0) hipLaunchKernelGGL((k_splat_bwd_view<D, false>), dim3(bx, ny), dim3(WG), 0, stream, d, params, s, rec, items, n2, fuse ? 1 : 0, samples, grad, none);
    if (x) hipLaunchKernelGGL(k_exchange_pack, dim3(bx), dim3(WG), 0, stream, d, s, samples, (const float4*)rec, *x, items, fuse ? n2 : 0);
    else if (fu && fu->params && s.mean_copy) {
        hipLaunchKernelGGL((k_splat_bwd_reduce<D, true>), dim3(bx, reduce_parts<D>()), dim3(WG), 0, stream, d, params, s, samples, (const float4*)rec, grad, items,
                           fuse ? n2 + n1 : 0, *fu);
        if (update_applied) *update_applied = true;
    } else hipLaunchKernelGGL((k_splat_bwd_reduce<D, false>), dim3(bx, reduce_parts<D>()), dim3(WG), 0, stream, d, params, s, samples, (const float4*)rec, grad, items,
                              fuse ? n2 + n1 : 0, FusedUpdate{});
}

int launch_splat_backward_avg(const Dims& d, const float* params, const Scratch& s, float samples, float* grad, const int* items,
                              int n_pairs, int n_singles, bool fuse_pairs, hipStream_t stream, const Exchange* x, const FusedUpdate* fu,
                              bool* update_applied) {
    if (update_applied) *update_applied = false;
    if (d.P == 0) return GS_OK;
    switch (d.D) {
        case 0: launch_splat_avg_d<0>(d, params, s, samples, grad, items, n_pairs, n_singles, fuse_pairs, x, stream, fu, update_applied); break;
        case 1: launch_splat_avg_d<1>(d, params, s, samples, grad, items, n_pairs, n_singles, fuse_pairs, x, stream, fu, update_applied); break;
        case 2: launch_splat_avg_d<2>(d, params, s, samples, grad, items, n_pairs, n_singles, fuse_pairs, x, stream, fu, update_applied); break;
        default: launch_splat_avg_d<3>(d, params, s, samples, grad, items, n_pairs, n_singles, fuse_pairs, x, stream, fu, update_applied); break;
    }
    GS_HIP(hipGetLastError());
    return GS_OK;
}

template <int D>
static void launch_sh_rebuild_d(const Dims& d, const float* params, const float* mean_copy, const Exchange& x, int n_cameras, int pp, float samples, float* grad,
                                int parts, const FusedUpdate* fu, hipStream_t stream) {
    const int first = (parts & 2) ? 0 : 1;
    const int count = ((parts & 2) ? 1 : 0) + ((parts & 1) ? reduce_parts<D>() - 1 : 0);
    const dim3 grid((d.P + WG - 1) / WG, count);
    if (fu && fu->params && mean_copy)
        hipLaunchKernelGGL((k_sh_rebuild<D, true>), grid, dim3(WG), 0, stream, d, params, mean_copy, x, n_cameras, pp, samples, grad, first, *fu);
    else
        hipLaunchKernelGGL((k_sh_rebuild<D, false>), grid, dim3(WG), 0, stream, d, params, mean_copy, x, n_cameras, pp, samples, grad, first, FusedUpdate{});
}
int launch_sh_rebuild(const Dims& d, const float* params, const float* mean_copy, const Exchange& x, int n_cameras, bool per_pass, float samples,
                      float* grad, int parts, hipStream_t stream, const FusedUpdate* fu) {
    if (d.P == 0 || !(parts & 3)) return GS_OK;
    const int pp = per_pass ? 1 : 0;
    switch (d.D) {
        case 0: launch_sh_rebuild_d<0>(d, params, mean_copy, x, n_cameras, pp, samples, grad, parts, fu, stream); break;
        case 1: launch_sh_rebuild_d<1>(d, params, mean_copy, x, n_cameras, pp, samples, grad, parts, fu, stream); break;
        case 2: launch_sh_rebuild_d<2>(d, params, mean_copy, x, n_cameras, pp, samples, grad, parts, fu, stream); break;
        default: launch_sh_rebuild_d<3>(d, params, mean_copy, x, n_cameras, pp, samples, grad, parts, fu, stream); break;
    }
    GS_HIP(hipGetLastError());
    return GS_OK;
}

// Seam form: view 0 only, reference-shaped AoS outputs.  Pixel-stage buffers are accumulated into
// (the reference's atomicAdd targets), the per-splat results are assigned, culled splats untouched.
template <int D>
__global__ __launch_bounds__(WG) void k_splat_bwd_seam(Dims d, const float* __restrict__ params, Scratch s, SeamGrads g) {
    constexpr int NC = (D + 1) * (D + 1);
    const int i = blockIdx.x * WG + threadIdx.x;
    if (i >= d.P) return;
    const GeomRec* rec = s.geom + i;
    if (!(rec->radius > 0)) return;
    const Planes pl{ d.M };
    const size_t st = (size_t)d.Pa;
    float mean[3], sc[3], q[4], sh[NC][3];
#pragma unroll
    for (int c = 0; c < 3; c++) { mean[c] = params[pl.loc(c) * st + i]; sc[c] = d.mod * params[pl.scale(c) * st + i]; }
#pragma unroll
    for (int c = 0; c < 4; c++) q[c] = params[pl.rot(c) * st + i];
#pragma unroll
    for (int k = 0; k < NC; k++)
#pragma unroll
        for (int c = 0; c < 3; c++) sh[k][c] = params[pl.sh(k, c) * st + i];
    const uint32_t tiles = s.tiles_touched[i];
    const uint32_t first = s.point_offsets[i] - tiles;
    float sum[9];
    gather_rows(s.G, s.row_epoch, uses_row_marks(d, s, 0) ? (uint32_t)d.epoch : 0u, s.zero_row, first, tiles, sum);
    float dcolor[3];
    for (int c = 0; c < 3; c++) { dcolor[c] = g.dL_dcolor[3 * (size_t)i + c] + sum[c]; g.dL_dcolor[3 * (size_t)i + c] = dcolor[c]; }
    const float g2x = g.dL_dmean2D[3 * (size_t)i] + sum[3], g2y = g.dL_dmean2D[3 * (size_t)i + 1] + sum[4];
    g.dL_dmean2D[3 * (size_t)i] = g2x; g.dL_dmean2D[3 * (size_t)i + 1] = g2y;
    const float gcx = g.dL_dconic[4 * (size_t)i] + sum[5], gcy = g.dL_dconic[4 * (size_t)i + 1] + sum[6],
                gcz = g.dL_dconic[4 * (size_t)i + 3] + sum[7];
    g.dL_dconic[4 * (size_t)i] = gcx; g.dL_dconic[4 * (size_t)i + 1] = gcy; g.dL_dconic[4 * (size_t)i + 3] = gcz;
    g.dL_dopacity[i] += sum[8];
    SplatOut<D> o;
    float dRGB[3];
    auto jac_at = [&](int ch, float X, float Y, float Z, float& dx_, float& dy_, float& dz_) {
        sh_direction_jacobian<D>(X, Y, Z, [&](int k) { return sh[k][ch]; }, dx_, dy_, dz_);
    };
    splat_backward_core<D, true>(s.views[0], d.W, d.H, mean, sc, q, jac_at, rec->flags, dcolor, g2x, g2y, gcx, gcy, gcz, o, dRGB);
    for (int c = 0; c < 3; c++) { g.dL_dmean3D[3 * (size_t)i + c] = o.mean[c]; g.dL_dscale[3 * (size_t)i + c] = o.scale[c]; }
    for (int c = 0; c < 6; c++) g.dL_dcov3D[6 * (size_t)i + c] = o.cov3D[c];
    for (int c = 0; c < 4; c++) g.dL_drot[4 * (size_t)i + c] = o.rot[c];
#pragma unroll
    for (int k = 0; k < NC; k++)
#pragma unroll
        for (int c = 0; c < 3; c++) g.dL_dsh[((size_t)i * d.M + k) * 3 + c] = o.sh[k][c];
}

int launch_splat_backward_seam(const Dims& d, const float* params, const Scratch& s, const SeamGrads& g, hipStream_t stream) {
    if (d.P == 0) return GS_OK;
    dim3 grid((d.P + WG - 1) / WG);
    switch (d.D) {
        case 0: hipLaunchKernelGGL(k_splat_bwd_seam<0>, grid, dim3(WG), 0, stream, d, params, s, g); break;
        case 1: hipLaunchKernelGGL(k_splat_bwd_seam<1>, grid, dim3(WG), 0, stream, d, params, s, g); break;
        case 2: hipLaunchKernelGGL(k_splat_bwd_seam<2>, grid, dim3(WG), 0, stream, d, params, s, g); break;
        default: hipLaunchKernelGGL(k_splat_bwd_seam<3>, grid, dim3(WG), 0, stream, d, params, s, g); break;
    }
    GS_HIP(hipGetLastError());
    return GS_OK;
}

}  // namespace gs
