// gs_oracle.cpp — CPU restatement of the reference training step.  TEST INFRASTRUCTURE ONLY.
//
// Nothing outside tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this
// library; the shipped HIP path never links or loads it.
//
// PARITY UNPINNED: the reference (osreboot/Gaussian-Splatterer v1.1.0) ships no tests, golden
// vectors or fixtures, it cannot be compiled here (CUDA + five empty submodules), and the
// rasterizer arithmetic lives in the un-vendored, un-pinned submodule
// graphdeco-inria/diff-gaussian-rasterization (.gitmodules:10-12).  This file restates
//   * the reference's own kernels and step orchestration  (src/Trainer.cu, cited per function), and
//   * the published algorithm of that rasterizer at the API era fixed by the reference's call
//     sites (src/Trainer.cu:175-201, :334-360, :378-412),
// and is pinned only by source-independent checks: fp64 finite differences of the forward
// against the analytic backward, closed-form known-answer cases and structural invariants
// (tests/test_oracle_*.py).
//
// Arithmetic: every function is a template over Real.  Real=float is the oracle proper (fp32,
// built with -ffp-contract=off so that no multiply-add is fused); Real=double is used only to
// validate the analytic backward by finite differences.
//
// Matrices are glm column-major float[16]: m[col*4+row]  (glm::value_ptr, src/Trainer.cu:323-324).

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <set>
#include <string>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

constexpr int TILE = 16;  // BLOCK_X = BLOCK_Y = 16

// ---------------------------------------------------------------------------------------------
// small helpers (SURVEY Appendix A "Helpers")
// ---------------------------------------------------------------------------------------------
template <class R> struct V3 { R x, y, z; };
template <class R> struct V4 { R x, y, z, w; };

template <class R> inline V3<R> transformPoint4x3(const V3<R>& p, const R* m) {
    return { m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
             m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
             m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14] };
}
template <class R> inline V4<R> transformPoint4x4(const V3<R>& p, const R* m) {
    return { m[0] * p.x + m[4] * p.y + m[8] * p.z + m[12],
             m[1] * p.x + m[5] * p.y + m[9] * p.z + m[13],
             m[2] * p.x + m[6] * p.y + m[10] * p.z + m[14],
             m[3] * p.x + m[7] * p.y + m[11] * p.z + m[15] };
}
template <class R> inline V3<R> transformVec4x3Transpose(const V3<R>& p, const R* m) {
    return { m[0] * p.x + m[1] * p.y + m[2] * p.z,
             m[4] * p.x + m[5] * p.y + m[6] * p.z,
             m[8] * p.x + m[9] * p.y + m[10] * p.z };
}
template <class R> inline R ndc2Pix(R v, int S) { return ((v + R(1.0)) * R(S) - R(1.0)) * R(0.5); }

template <class R> struct SH {
    static constexpr R C0 = R(0.28209479177387814);
    static constexpr R C1 = R(0.4886025119029199);
    static constexpr R C2[5] = { R(1.0925484305920792), R(-1.0925484305920792), R(0.31539156525252005),
                                 R(-1.0925484305920792), R(0.5462742152960396) };
    static constexpr R C3[7] = { R(-0.5900435899266435), R(2.890611442640554), R(-0.4570457994644658),
                                 R(0.3731763325901154), R(-0.4570457994644658), R(1.445305721320277),
                                 R(-0.5900435899266435) };
};

inline uint32_t depth_bits(float d) { uint32_t u; std::memcpy(&u, &d, 4); return u; }

// ---------------------------------------------------------------------------------------------
// Per-view state produced by forward and consumed by backward (the reference's geometry /
// binning / image scratch chunks, src/Trainer.cu:329-337,399-401).
// ---------------------------------------------------------------------------------------------
template <class R> struct ViewState {
    int P = 0, W = 0, H = 0, gx = 0, gy = 0;
    std::vector<R> depth, means2D, cov3D, conic_opacity, rgb;
    std::vector<int32_t> radii;
    std::vector<uint32_t> tiles_touched, point_offsets, rect;  // rect: minx,miny,maxx,maxy per splat
    std::vector<uint8_t> clamped;
    std::vector<uint64_t> keys_unsorted, keys;
    std::vector<uint32_t> vals_unsorted, point_list;
    std::vector<uint32_t> ranges;  // 2 per tile
    std::vector<R> final_T;
    std::vector<uint32_t> n_contrib;
    std::vector<float> margin;  // per-pixel fragility: min relative distance to a discrete threshold
    std::vector<float> pmargin; // per-pixel: min over its pairs of |power| / (2^-24 m), m = the magnitudes the exponent's products cancel from: how
                                // many ulps of its own conditioning the `power > 0: skip` decision of the most fragile pair is away from flipping
    std::vector<float> splat_margin;  // per-splat fragility: min margin over the pixels the splat is blended at
    int R_total = 0;
};

// A.1 — per-splat preprocess.  Operation order follows glm's mat3 product (column-by-column,
// k = 0,1,2 summed left to right) so that the HIP kernel can reproduce every bit.
template <class R>
void preprocess(int P, int D, int M, const R* means, const R* scales, R mod, const R* rots, const R* opac,
                const R* shs, const R* view, const R* proj, const R* campos, int W, int H, R tanx, R tany,
                ViewState<R>& g) {
    g.P = P; g.W = W; g.H = H;
    g.gx = (W + TILE - 1) / TILE; g.gy = (H + TILE - 1) / TILE;
    g.depth.assign(P, 0); g.means2D.assign(2 * (size_t)P, 0); g.cov3D.assign(6 * (size_t)P, 0);
    g.conic_opacity.assign(4 * (size_t)P, 0); g.rgb.assign(3 * (size_t)P, 0);
    g.radii.assign(P, 0); g.tiles_touched.assign(P, 0); g.rect.assign(4 * (size_t)P, 0);
    g.clamped.assign(3 * (size_t)P, 0);
    const R focal_x = R(W) / (R(2.0) * tanx);
    const R focal_y = R(H) / (R(2.0) * tany);

#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        const V3<R> p = { means[3 * i], means[3 * i + 1], means[3 * i + 2] };
        // frustum test: only the near plane is active upstream
        const V3<R> pv = transformPoint4x3(p, view);
        if (pv.z <= R(0.2)) continue;
        const V4<R> ph = transformPoint4x4(p, proj);
        const R p_w = R(1.0) / (ph.w + R(0.0000001));
        const R ppx = ph.x * p_w, ppy = ph.y * p_w;

        // cov3D = (S Rg)^T (S Rg) with Rg filled column-wise, quaternion (r,x,y,z) NOT normalised
        const R sx = mod * scales[3 * i], sy = mod * scales[3 * i + 1], sz = mod * scales[3 * i + 2];
        const R r = rots[4 * i], x = rots[4 * i + 1], y = rots[4 * i + 2], z = rots[4 * i + 3];
        R Rg[3][3];  // [col][row]
        Rg[0][0] = R(1.0) - R(2.0) * (y * y + z * z); Rg[0][1] = R(2.0) * (x * y - r * z); Rg[0][2] = R(2.0) * (x * z + r * y);
        Rg[1][0] = R(2.0) * (x * y + r * z); Rg[1][1] = R(1.0) - R(2.0) * (x * x + z * z); Rg[1][2] = R(2.0) * (y * z - r * x);
        Rg[2][0] = R(2.0) * (x * z - r * y); Rg[2][1] = R(2.0) * (y * z + r * x); Rg[2][2] = R(1.0) - R(2.0) * (x * x + y * y);
        R Mm[3][3];  // M = S * Rg : M[c][k] = s_k * Rg[c][k]
        for (int c = 0; c < 3; c++) { Mm[c][0] = sx * Rg[c][0]; Mm[c][1] = sy * Rg[c][1]; Mm[c][2] = sz * Rg[c][2]; }
        auto sig = [&](int c, int rr) { return Mm[rr][0] * Mm[c][0] + Mm[rr][1] * Mm[c][1] + Mm[rr][2] * Mm[c][2]; };
        R c3[6] = { sig(0, 0), sig(0, 1), sig(0, 2), sig(1, 1), sig(1, 2), sig(2, 2) };
        for (int k = 0; k < 6; k++) g.cov3D[6 * (size_t)i + k] = c3[k];

        // cov2D (EWA): T = W*J (glm), cov = T^T Vrk^T T, +0.3 low-pass on the diagonal
        V3<R> t = pv;
        const R limx = R(1.3) * tanx, limy = R(1.3) * tany;
        const R txtz = t.x / t.z, tytz = t.y / t.z;
        t.x = std::min(limx, std::max(-limx, txtz)) * t.z;
        t.y = std::min(limy, std::max(-limy, tytz)) * t.z;
        const R J00 = focal_x / t.z, J02 = -(focal_x * t.x) / (t.z * t.z);
        const R J11 = focal_y / t.z, J12 = -(focal_y * t.y) / (t.z * t.z);
        R T[2][3];  // T[c][r], c = image axis, r = world axis;  W[k][r] = view[4r+k]
        for (int rr = 0; rr < 3; rr++) {
            T[0][rr] = view[4 * rr] * J00 + view[4 * rr + 2] * J02;
            T[1][rr] = view[4 * rr + 1] * J11 + view[4 * rr + 2] * J12;
        }
        const R V[3][3] = { { c3[0], c3[1], c3[2] }, { c3[1], c3[3], c3[4] }, { c3[2], c3[4], c3[5] } };
        R A[3][2];  // A[k][r] = sum_l T[r][l] * V[l][k]
        for (int k = 0; k < 3; k++)
            for (int rr = 0; rr < 2; rr++) A[k][rr] = T[rr][0] * V[0][k] + T[rr][1] * V[1][k] + T[rr][2] * V[2][k];
        R ca = A[0][0] * T[0][0] + A[1][0] * T[0][1] + A[2][0] * T[0][2];
        const R cb = A[0][1] * T[0][0] + A[1][1] * T[0][1] + A[2][1] * T[0][2];
        R cc = A[0][1] * T[1][0] + A[1][1] * T[1][1] + A[2][1] * T[1][2];
        ca += R(0.3); cc += R(0.3);

        const R det = ca * cc - cb * cb;
        if (det == R(0.0)) continue;
        const R det_inv = R(1.0) / det;
        const R conx = cc * det_inv, cony = -cb * det_inv, conz = ca * det_inv;
        const R mid = R(0.5) * (ca + cc);
        const R lambda1 = mid + std::sqrt(std::max(R(0.1), mid * mid - det));
        const R lambda2 = mid - std::sqrt(std::max(R(0.1), mid * mid - det));
        const R my_radius = std::ceil(R(3.0) * std::sqrt(std::max(lambda1, lambda2)));
        const R px = ndc2Pix(ppx, W), py = ndc2Pix(ppy, H);
        const int max_radius = (int)my_radius;
        const int rminx = std::min(g.gx, std::max(0, (int)((px - R(max_radius)) / R(TILE))));
        const int rminy = std::min(g.gy, std::max(0, (int)((py - R(max_radius)) / R(TILE))));
        const int rmaxx = std::min(g.gx, std::max(0, (int)((px + R(max_radius) + R(TILE - 1)) / R(TILE))));
        const int rmaxy = std::min(g.gy, std::max(0, (int)((py + R(max_radius) + R(TILE - 1)) / R(TILE))));
        if ((rmaxx - rminx) * (rmaxy - rminy) == 0) continue;

        // colour from SH (colors_precomp == nullptr, src/Trainer.cu:346)
        V3<R> dir = { p.x - campos[0], p.y - campos[1], p.z - campos[2] };
        const R len = std::sqrt(dir.x * dir.x + dir.y * dir.y + dir.z * dir.z);
        dir.x = dir.x / len; dir.y = dir.y / len; dir.z = dir.z / len;
        const R* sh = shs + (size_t)i * M * 3;
        R res[3];
        for (int c = 0; c < 3; c++) {
            R v = SH<R>::C0 * sh[c];
            if (D > 0) {
                const R X = dir.x, Y = dir.y, Z = dir.z;
                v = v - SH<R>::C1 * Y * sh[3 + c] + SH<R>::C1 * Z * sh[6 + c] - SH<R>::C1 * X * sh[9 + c];
                if (D > 1) {
                    const R xx = X * X, yy = Y * Y, zz = Z * Z, xy = X * Y, yz = Y * Z, xz = X * Z;
                    v = v + SH<R>::C2[0] * xy * sh[12 + c] + SH<R>::C2[1] * yz * sh[15 + c] +
                        SH<R>::C2[2] * (R(2.0) * zz - xx - yy) * sh[18 + c] + SH<R>::C2[3] * xz * sh[21 + c] +
                        SH<R>::C2[4] * (xx - yy) * sh[24 + c];
                    if (D > 2) {
                        v = v + SH<R>::C3[0] * Y * (R(3.0) * xx - yy) * sh[27 + c] + SH<R>::C3[1] * xy * Z * sh[30 + c] +
                            SH<R>::C3[2] * Y * (R(4.0) * zz - xx - yy) * sh[33 + c] +
                            SH<R>::C3[3] * Z * (R(2.0) * zz - R(3.0) * xx - R(3.0) * yy) * sh[36 + c] +
                            SH<R>::C3[4] * X * (R(4.0) * zz - xx - yy) * sh[39 + c] + SH<R>::C3[5] * Z * (xx - yy) * sh[42 + c] +
                            SH<R>::C3[6] * X * (xx - R(3.0) * yy) * sh[45 + c];
                    }
                }
            }
            v += R(0.5);
            g.clamped[3 * (size_t)i + c] = (v < R(0.0));
            res[c] = std::max(v, R(0.0));
        }
        for (int c = 0; c < 3; c++) g.rgb[3 * (size_t)i + c] = res[c];
        g.depth[i] = pv.z;
        g.radii[i] = (int)my_radius;
        g.means2D[2 * (size_t)i] = px; g.means2D[2 * (size_t)i + 1] = py;
        g.conic_opacity[4 * (size_t)i] = conx; g.conic_opacity[4 * (size_t)i + 1] = cony;
        g.conic_opacity[4 * (size_t)i + 2] = conz; g.conic_opacity[4 * (size_t)i + 3] = opac[i];
        g.rect[4 * (size_t)i] = rminx; g.rect[4 * (size_t)i + 1] = rminy;
        g.rect[4 * (size_t)i + 2] = rmaxx; g.rect[4 * (size_t)i + 3] = rmaxy;
        g.tiles_touched[i] = (uint32_t)((rmaxy - rminy) * (rmaxx - rminx));
    }
}

// A.2-A.5 — inclusive scan, key duplication, stable sort by (tile<<32 | depth bits), tile ranges.
template <class R> void bin_and_sort(ViewState<R>& g) {
    const int P = g.P;
    g.point_offsets.assign(P, 0);
    uint32_t run = 0;
    for (int i = 0; i < P; i++) { run += g.tiles_touched[i]; g.point_offsets[i] = run; }
    const size_t Rn = run;
    g.R_total = (int)Rn;
    g.keys_unsorted.assign(Rn, 0); g.vals_unsorted.assign(Rn, 0);
    for (int i = 0; i < P; i++) {
        if (g.radii[i] <= 0) continue;
        uint32_t off = (i == 0) ? 0 : g.point_offsets[i - 1];
        const uint32_t* rc = &g.rect[4 * (size_t)i];
        for (uint32_t y = rc[1]; y < rc[3]; y++)
            for (uint32_t x = rc[0]; x < rc[2]; x++) {
                uint64_t key = (uint64_t)(y * (uint32_t)g.gx + x);
                key <<= 32;
                key |= depth_bits((float)g.depth[i]);
                g.keys_unsorted[off] = key; g.vals_unsorted[off] = (uint32_t)i; off++;
            }
    }
    std::vector<uint32_t> perm(Rn);
    for (size_t k = 0; k < Rn; k++) perm[k] = (uint32_t)k;
    std::stable_sort(perm.begin(), perm.end(),
                     [&](uint32_t a, uint32_t b) { return g.keys_unsorted[a] < g.keys_unsorted[b]; });
    g.keys.resize(Rn); g.point_list.resize(Rn);
    for (size_t k = 0; k < Rn; k++) { g.keys[k] = g.keys_unsorted[perm[k]]; g.point_list[k] = g.vals_unsorted[perm[k]]; }
    const int T = g.gx * g.gy;
    g.ranges.assign(2 * (size_t)T, 0);
    for (size_t k = 0; k < Rn; k++) {
        const uint32_t tile = (uint32_t)(g.keys[k] >> 32);
        if (k == 0) g.ranges[2 * (size_t)tile] = 0;
        else {
            const uint32_t prev = (uint32_t)(g.keys[k - 1] >> 32);
            if (prev != tile) { g.ranges[2 * (size_t)prev + 1] = (uint32_t)k; g.ranges[2 * (size_t)tile] = (uint32_t)k; }
        }
        if (k == Rn - 1) g.ranges[2 * (size_t)tile + 1] = (uint32_t)Rn;
    }
}

// |power| in units of the exponent's own conditioning, 2^-24 (|a| dx^2 / 2 + |c| dy^2 / 2 + |b dx dy|): below a few of them the sign of the
// computed power — the `power > 0: skip` decision — belongs to the evaluation order, not to the scene (pixel_run, PixelFlip kind 3).
template <class R> inline double power_magnitude(const R* co, R dx, R dy) {
    return 0.5 * (std::fabs((double)co[0]) * dx * dx + std::fabs((double)co[2]) * dy * dy) + std::fabs((double)co[1] * dx * dy);
}
template <class R> inline float power_ulps_from_zero(const R* co, R dx, R dy, R power) {
    const double m = power_magnitude<R>(co, dx, dy);
    return m > 0.0 ? (float)std::min(1e30, std::fabs((double)power) / (5.9604644775390625e-08 * m)) : 1e30f;
}
// The other two decisions see the same conditioning.  alpha carries m ulps of relative error (m = power_magnitude; none once clipped to
// 0.99), so its distance from 1/255 is only meaningful beyond that; T in front of an entry is a product of n factors (1 - alpha_i), each
// of which carries m_i alpha_i / (1 - alpha_i) ulps plus one for the multiplication: `tcond` below.  A decision is FRAGILE when its relative
// distance from the threshold is below  margin + ulps x 2^-24 x (its conditioning)  — ulps = exp_cond (pixel check) / power_ulps (flip
// allowance); with ulps = 0 this is the plain margin of rounds 3-4.  Found by running the sweep on other seeds: five of 352 scenes had one
// pixel each where a needle's alpha sat 1.3e-4 below 1/255 for the oracle and above it for the GPU (margin 1e-4).
constexpr double ULP24 = 5.9604644775390625e-08;

// A.6 — front-to-back alpha compositing, one pixel at a time, integer pixel centres.
template <class R> void render_forward(ViewState<R>& g, const R* bg, R* out_color) {
    const int W = g.W, H = g.H;
    const size_t N = (size_t)W * H;
    g.final_T.assign(N, 0); g.n_contrib.assign(N, 0); g.margin.assign(N, 1.0f); g.pmargin.assign(N, 1e30f);
    g.splat_margin.assign(g.P, 1.0f);
    // lock-free min on non-negative floats (their bit patterns order like unsigned integers); tiles run in parallel
    auto splat_min = [&](uint32_t id, float m) {
        uint32_t* p = reinterpret_cast<uint32_t*>(&g.splat_margin[id]);
        uint32_t nb; std::memcpy(&nb, &m, 4);
        uint32_t cur = __atomic_load_n(p, __ATOMIC_RELAXED);
        while (nb < cur && !__atomic_compare_exchange_n(p, &cur, nb, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    };
    const int T = g.gx * g.gy;
#pragma omp parallel for schedule(dynamic, 4)
    for (int tile = 0; tile < T; tile++) {
        const int tx = tile % g.gx, ty = tile / g.gx;
        const uint32_t beg = g.ranges[2 * (size_t)tile], end = g.ranges[2 * (size_t)tile + 1];
        for (int py = ty * TILE; py < std::min(H, (ty + 1) * TILE); py++)
            for (int px = tx * TILE; px < std::min(W, (tx + 1) * TILE); px++) {
                const R pixfx = (R)px, pixfy = (R)py;
                R Tt = R(1.0), C[3] = { 0, 0, 0 };
                uint32_t contributor = 0, last = 0;
                float marg = 1.0f, pmarg = 1e30f;
                double tcond = 0.0;  // conditioning of T so far, in ulps
                for (uint32_t k = beg; k < end; k++) {
                    contributor++;
                    const uint32_t id = g.point_list[k];
                    const R dx = g.means2D[2 * (size_t)id] - pixfx, dy = g.means2D[2 * (size_t)id + 1] - pixfy;
                    const R* co = &g.conic_opacity[4 * (size_t)id];
                    const R power = R(-0.5) * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    pmarg = std::min(pmarg, power_ulps_from_zero<R>(co, dx, dy, power));
                    if (power > R(0.0)) continue;
                    const R alpha = std::min(R(0.99), co[3] * std::exp(power));
                    const float ma = (float)(std::fabs(alpha - R(1.0 / 255.0)) * R(255.0));
                    marg = std::min(marg, ma);
                    const double m = alpha < R(0.99) ? power_magnitude<R>(co, dx, dy) : 0.0;
                    if (m > 0.0) pmarg = std::min(pmarg, (float)std::min(1e30, (double)ma / (ULP24 * m)));
                    if (alpha < R(1.0) / R(255.0)) continue;
                    const R test_T = Tt * (R(1.0) - alpha);
                    const float mt = (float)(std::fabs(test_T - R(0.0001)) * R(10000.0));
                    marg = std::min(marg, mt);
                    tcond += m * (double)alpha / (1.0 - (double)alpha) + 1.0;
                    pmarg = std::min(pmarg, (float)std::min(1e30, (double)mt / (ULP24 * tcond)));
                    if (test_T < R(0.0001)) break;  // done: this entry is NOT applied
                    for (int c = 0; c < 3; c++) C[c] += g.rgb[3 * (size_t)id + c] * alpha * Tt;
                    Tt = test_T;
                    last = contributor;
                }
                const size_t pix = (size_t)py * W + px;
                g.final_T[pix] = Tt; g.n_contrib[pix] = last; g.margin[pix] = marg; g.pmargin[pix] = pmarg;
                // A flipped decision at this pixel moves T / the colour recurrence of EVERY splat blended here, not
                // only the one that flipped: the pixel's fragility is inherited by all entries that (nearly) reach it.
                if (marg < 1e-2f)
                    for (uint32_t k = beg; k < end; k++) {
                        const uint32_t id = g.point_list[k];
                        const R dx = g.means2D[2 * (size_t)id] - pixfx, dy = g.means2D[2 * (size_t)id + 1] - pixfy;
                        const R* co = &g.conic_opacity[4 * (size_t)id];
                        const R power = R(-0.5) * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                        if (power <= R(0.0) && co[3] * std::exp(power) >= R(0.99 / 255.0)) splat_min(id, marg);
                    }
                for (int c = 0; c < 3; c++) out_color[c * N + pix] = C[c] + Tt * bg[c];
            }
    }
}

// Per-splat outputs of the backward pass, reference buffer shapes (src/Trainer.cu:291-300).
template <class R> struct Grads {
    R *dL_dmean2D, *dL_dconic, *dL_dopacity, *dL_dcolor, *dL_dmean3D, *dL_dcov3D, *dL_dsh, *dL_dscale, *dL_drot;
};

// A.7 — reverse traversal.  The reference sums the per-pixel terms with float atomicAdd in
// arbitrary order; here each (tile,entry) partial is summed in double in pixel order and the
// partials are then added per splat in sorted-list order, i.e. the correctly rounded sum of the
// same fp32 terms.  `abs9` (optional) receives sum|term| per splat for tolerance floors.
//
// Decision flips.  The blend has two discrete decisions per (pixel, entry) pair: "alpha < 1/255 -> skip" and
// "T(1-alpha) < 1e-4 -> stop, entry not applied".  An implementation whose exp() differs from libm's in the last
// bit may take the other branch for a pair that sits on a threshold, which moves the terms of EVERY splat blended
// at that pixel.  PixelFlip forces one such decision the other way; with `flip9` (optional) render_backward re-runs
// each pixel once per pair that lies within `flip_margin` (relative) of a threshold with that decision inverted and
// accumulates |term_flipped - term| per splat and sum: the admissible deviation of an implementation that flips
// those pairs.  The parity tests use  tol = 1e-4 * sum|term| + flip9(margin)  and report how the count of
// out-of-tolerance splats depends on the margin (tests/test_gpu_raster.py).
struct PixelFlip { long k = -1; int kind = 0; };  // kind 1: alpha test of list entry k inverted, 2: T test inverted, 3: `power > 0: skip` inverted

// Forward blend of one pixel over list entries [beg, end) -> (final T, last contributor), as render_forward does,
// with an optional forced decision.  Collects the fragile pairs when `frag` is given.
template <class R>
inline void pixel_forward(const ViewState<R>& g, uint32_t beg, uint32_t end, R pixfx, R pixfy, PixelFlip flip, R& T_out,
                          uint32_t& last_out, std::vector<PixelFlip>* frag, float frag_margin, float power_ulps = 0.0f) {
    R Tt = R(1.0);
    uint32_t contributor = 0, last = 0;
    double tcond = 0.0;
    for (uint32_t k = beg; k < end; k++) {
        contributor++;
        const uint32_t id = g.point_list[k];
        const R dx = g.means2D[2 * (size_t)id] - pixfx, dy = g.means2D[2 * (size_t)id + 1] - pixfy;
        const R* co = &g.conic_opacity[4 * (size_t)id];
        const R power = R(-0.5) * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
        bool pskip = power > R(0.0);
        if (frag && power_ulps > 0.0f && power_ulps_from_zero<R>(co, dx, dy, power) < power_ulps) frag->push_back({ (long)k, 3 });
        if (flip.kind == 3 && flip.k == (long)k) pskip = !pskip;
        if (pskip) continue;
        const R alpha = std::min(R(0.99), co[3] * std::exp(power));
        const double m = (power_ulps > 0.0f && alpha < R(0.99)) ? power_magnitude<R>(co, dx, dy) : 0.0;
        bool skip = alpha < R(1.0) / R(255.0);
        if (frag && (double)(std::fabs(alpha - R(1.0 / 255.0)) * R(255.0)) < (double)frag_margin + power_ulps * ULP24 * m) frag->push_back({ (long)k, 1 });
        if (flip.kind == 1 && flip.k == (long)k) skip = !skip;
        if (skip) continue;
        const R test_T = Tt * (R(1.0) - alpha);
        bool stop = test_T < R(0.0001);
        tcond += m * (double)alpha / (1.0 - (double)alpha) + 1.0;
        if (frag && (double)(std::fabs(test_T - R(0.0001)) * R(10000.0)) < (double)frag_margin + power_ulps * ULP24 * tcond) frag->push_back({ (long)k, 2 });
        if (flip.kind == 2 && flip.k == (long)k) stop = !stop;
        if (stop) break;
        Tt = test_T;
        last = contributor;
    }
    T_out = Tt; last_out = last;
}

// Reverse traversal of one pixel (upstream renderCUDA backward); emit(k, q, term) receives the nine terms of every
// blended entry.  The alpha test honours the same forced decision as pixel_forward.
template <class R, class Emit>
inline void pixel_backward(const ViewState<R>& g, uint32_t beg, uint32_t end, R pixfx, R pixfy, R T_final, uint32_t last_contributor,
                           const R* bg, const R dpx[3], R ddelx_dx, R ddely_dy, PixelFlip flip, Emit emit) {
    R Tt = T_final;
    R accum_rec[3] = { 0, 0, 0 }, last_color[3] = { 0, 0, 0 }, last_alpha = 0;
    uint32_t contributor = end - beg;
    for (uint32_t k = end; k-- > beg;) {
        contributor--;
        if (contributor >= last_contributor) continue;
        const uint32_t id = g.point_list[k];
        const R dx = g.means2D[2 * (size_t)id] - pixfx, dy = g.means2D[2 * (size_t)id + 1] - pixfy;
        const R* co = &g.conic_opacity[4 * (size_t)id];
        const R power = R(-0.5) * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
        bool pskip = power > R(0.0);
        if (flip.kind == 3 && flip.k == (long)k) pskip = !pskip;
        if (pskip) continue;
        const R G = std::exp(power);
        const R alpha = std::min(R(0.99), co[3] * G);
        bool skip = alpha < R(1.0) / R(255.0);
        if (flip.kind == 1 && flip.k == (long)k) skip = !skip;
        if (skip) continue;
        Tt = Tt / (R(1.0) - alpha);
        const R dchannel_dcolor = alpha * Tt;
        R dL_dalpha = 0;
        for (int c = 0; c < 3; c++) {
            const R col = g.rgb[3 * (size_t)id + c];
            accum_rec[c] = last_alpha * last_color[c] + (R(1.0) - last_alpha) * accum_rec[c];
            last_color[c] = col;
            dL_dalpha += (col - accum_rec[c]) * dpx[c];
            emit(k, c, dchannel_dcolor * dpx[c]);
        }
        dL_dalpha *= Tt;
        last_alpha = alpha;
        R bg_dot = 0;
        for (int c = 0; c < 3; c++) bg_dot += bg[c] * dpx[c];
        dL_dalpha += (-T_final / (R(1.0) - alpha)) * bg_dot;
        const R dL_dG = co[3] * dL_dalpha;
        const R gdx = G * dx, gdy = G * dy;
        const R dG_ddelx = -gdx * co[0] - gdy * co[1];
        const R dG_ddely = -gdy * co[2] - gdx * co[1];
        emit(k, 3, dL_dG * dG_ddelx * ddelx_dx);
        emit(k, 4, dL_dG * dG_ddely * ddely_dy);
        emit(k, 5, R(-0.5) * gdx * dx * dL_dG);
        emit(k, 6, R(-0.5) * gdx * dy * dL_dG);
        emit(k, 7, R(-0.5) * gdy * dy * dL_dG);
        emit(k, 8, G * dL_dalpha);
    }
}

// Conditioning of one pixel's nine terms, in units of 2^-24 (first-order forward error analysis, all in double).  Three places
// of the blend subtract products that can be orders of magnitude larger than their difference:
//   * the exponent  power = -(a dx^2 + c dy^2)/2 - b dx dy  (m = |a| dx^2/2 + |c| dy^2/2 + |b dx dy| >> |power| for a big splat seen far
//     along its long axis): ANY fp32 evaluation order, the reference's included, carries ~m ulps of relative error in G = exp(power) and
//     in alpha; through T = prod(1 - alpha) and the colour recurrence that error reaches the terms of every other entry of the pixel;
//   * dG/dmean = -G (a dx + b dy), -G (c dy + b dx): the two products cancel along the same axis;
//   * dL/dalpha = T sum_c (colour_c - accum_c) dL/dpixel_c - T_final / (1 - alpha) (bg . dL/dpixel);
//   * and T itself, a product of as many factors as the pixel blends entries: one ulp apiece.
// An implementation that applies per-splat constants after summing over pixels (a . sum(u dx) + b . sum(u dy) instead of
// sum(u (a dx + b dy))) has the same bound with a different draw, which 1e-4 of sum|term| — terms AFTER the cancellation — cannot
// see.  emit(k, q, c): |error of term q of entry k| <~ c x 2^-24; the tests allow a small multiple of it (cond9).
template <class R, class Emit>
inline void pixel_cond(const ViewState<R>& g, uint32_t beg, uint32_t end, R pixfx, R pixfy, R T_final, uint32_t last_contributor,
                       const R* bg, const R dpx[3], R ddelx_dx, R ddely_dy, PixelFlip flip, Emit emit) {
    struct E { uint32_t k; double dx, dy, G, alpha, m, rho, a, b, c, op, col[3]; };
    std::vector<E> ent;   // back to front, as the backward visits them
    uint32_t contributor = end - beg;
    for (uint32_t k = end; k-- > beg;) {
        contributor--;
        if (contributor >= last_contributor) continue;
        const uint32_t id = g.point_list[k];
        const R dx = g.means2D[2 * (size_t)id] - pixfx, dy = g.means2D[2 * (size_t)id + 1] - pixfy;
        const R* co = &g.conic_opacity[4 * (size_t)id];
        const R power = R(-0.5) * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
        bool pskip = power > R(0.0);
        if (flip.kind == 3 && flip.k == (long)k) pskip = !pskip;
        if (pskip) continue;
        const R G = std::exp(power);
        const R alpha = std::min(R(0.99), co[3] * G);
        bool skip = alpha < R(1.0) / R(255.0);
        if (flip.kind == 1 && flip.k == (long)k) skip = !skip;
        if (skip) continue;
        const double m = 0.5 * (std::fabs((double)co[0]) * dx * dx + std::fabs((double)co[2]) * dy * dy) + std::fabs((double)co[1] * dx * dy);
        ent.push_back({ k, (double)dx, (double)dy, (double)G, (double)alpha, m, alpha < R(0.99) ? m : 0.0, (double)co[0], (double)co[1], (double)co[2],
                        (double)co[3], { (double)g.rgb[3 * (size_t)id], (double)g.rgb[3 * (size_t)id + 1], (double)g.rgb[3 * (size_t)id + 2] } });
    }
    const size_t n = ent.size();
    if (!n) return;
    // relative error of T in front of entry j: sum over the entries in front of it of rho alpha / (1 - alpha)
    std::vector<double> front(n, 0.0);
    double run = 0.0;
    for (size_t j = n; j-- > 0;) { front[j] = run; run += ent[j].rho * ent[j].alpha / (1.0 - ent[j].alpha); }
    // ... plus one ulp per factor: T is a running product of n factors (the forward multiplies front to back, a backward that starts
    // from T_final divides back to front), so the T any entry sees carries up to n ulps whatever the order — a heap of 12 000 faint
    // splats on one tile moved every sum of a splat blended there by 3.8e-4 of its size
    for (size_t j = 0; j < n; j++) front[j] += (double)n;
    const double front_all = run + (double)n;
    double Tt = (double)T_final, accum[3] = { 0, 0, 0 }, dacc[3] = { 0, 0, 0 }, last_col[3] = { 0, 0, 0 }, last_alpha = 0, last_rho = 0;
    double bg_dot = 0;
    for (int c = 0; c < 3; c++) bg_dot += (double)bg[c] * (double)dpx[c];
    for (size_t j = 0; j < n; j++) {
        const E& e = ent[j];
        Tt = Tt / (1.0 - e.alpha);
        double dLda = 0, dLda_err = 0;
        for (int c = 0; c < 3; c++) {
            const double prev = accum[c];
            accum[c] = last_alpha * last_col[c] + (1.0 - last_alpha) * prev;
            dacc[c] = last_alpha * last_rho * std::fabs(last_col[c] - prev) + (1.0 - last_alpha) * dacc[c];
            last_col[c] = e.col[c];
            dLda += (e.col[c] - accum[c]) * (double)dpx[c];
            dLda_err += std::fabs((double)dpx[c]) * (std::fabs(e.col[c] - accum[c]) * (front[j] + 1.0) + dacc[c]);
            emit(e.k, c, std::fabs(e.alpha * Tt * (double)dpx[c]) * (e.rho + front[j]));
        }
        dLda *= Tt; dLda_err *= Tt;
        last_alpha = e.alpha; last_rho = e.rho;
        const double bgpart = (-(double)T_final / (1.0 - e.alpha)) * bg_dot;
        dLda += bgpart; dLda_err += std::fabs(bgpart) * (front_all + 1.0);
        const double through = std::fabs(dLda) * e.m + dLda_err;       // G carries m ulps whether or not alpha is clipped
        const double gdx = e.G * e.dx, gdy = e.G * e.dy;
        emit(e.k, 3, e.op * (std::fabs(gdx * e.a) + std::fabs(gdy * e.b)) * (double)ddelx_dx * (through + std::fabs(dLda)));
        emit(e.k, 4, e.op * (std::fabs(gdy * e.c) + std::fabs(gdx * e.b)) * (double)ddely_dy * (through + std::fabs(dLda)));
        emit(e.k, 5, e.op * 0.5 * std::fabs(gdx * e.dx) * through);
        emit(e.k, 6, e.op * 0.5 * std::fabs(gdx * e.dy) * through);
        emit(e.k, 7, e.op * 0.5 * std::fabs(gdy * e.dy) * through);
        emit(e.k, 8, e.G * through);
    }
}

template <class R>
void render_backward(const ViewState<R>& g, const R* bg, const R* dL_dpix, const Grads<R>& o, double* abs9, double* flip9 = nullptr,
                     float flip_margin = 0.0f, double* cond9 = nullptr, float power_ulps = 0.0f) {
    const int W = g.W, H = g.H, P = g.P;
    const size_t N = (size_t)W * H;
    const size_t Rn = g.point_list.size();
    std::vector<double> part(Rn * 9, 0.0), partabs(abs9 ? Rn * 9 : 0, 0.0), partflip(flip9 ? Rn * 9 : 0, 0.0), partcond(cond9 ? Rn * 9 : 0, 0.0);
    const int T = g.gx * g.gy;
    const R ddelx_dx = R(0.5) * R(W), ddely_dy = R(0.5) * R(H);
#pragma omp parallel for schedule(dynamic, 4)
    for (int tile = 0; tile < T; tile++) {
        const int tx = tile % g.gx, ty = tile / g.gx;
        const uint32_t beg = g.ranges[2 * (size_t)tile], end = g.ranges[2 * (size_t)tile + 1];
        if (end <= beg) continue;
        std::vector<PixelFlip> frag;
        std::vector<double> t0, t1;
        for (int py = ty * TILE; py < std::min(H, (ty + 1) * TILE); py++)
            for (int px = tx * TILE; px < std::min(W, (tx + 1) * TILE); px++) {
                const size_t pix = (size_t)py * W + px;
                const R pixfx = (R)px, pixfy = (R)py;
                const R dpx[3] = { dL_dpix[pix], dL_dpix[N + pix], dL_dpix[2 * N + pix] };
                pixel_backward<R>(g, beg, end, pixfx, pixfy, g.final_T[pix], g.n_contrib[pix], bg, dpx, ddelx_dx, ddely_dy, PixelFlip{},
                                  [&](uint32_t k, int q, R term) {
                                      part[(size_t)k * 9 + q] += (double)term;
                                      if (abs9) partabs[(size_t)k * 9 + q] += std::fabs((double)term);
                                  });
                if (cond9)
                    pixel_cond<R>(g, beg, end, pixfx, pixfy, g.final_T[pix], g.n_contrib[pix], bg, dpx, ddelx_dx, ddely_dy, PixelFlip{},
                                  [&](uint32_t k, int q, double c) { partcond[(size_t)k * 9 + q] += c; });
                // (d < margin + ulps c  is implied by neither  d < margin  nor  d < ulps c  alone, but by  d < 2 max of the two: the
                //  pixel is re-examined on the doubled criteria, pixel_forward then applies the exact one)
                if (!flip9 || !(g.margin[pix] < 2.0f * flip_margin || (power_ulps > 0.0f && g.pmargin[pix] < 2.0f * power_ulps))) continue;
                // this pixel holds at least one pair within flip_margin of a threshold: one re-run per such pair
                frag.clear();
                R Tq; uint32_t lastq;
                pixel_forward<R>(g, beg, end, pixfx, pixfy, PixelFlip{}, Tq, lastq, &frag, flip_margin, power_ulps);
                if (frag.empty()) continue;
                const size_t n = (size_t)(end - beg) * 9;
                t0.assign(n, 0.0);
                pixel_backward<R>(g, beg, end, pixfx, pixfy, Tq, lastq, bg, dpx, ddelx_dx, ddely_dy, PixelFlip{},
                                  [&](uint32_t k, int q, R term) { t0[(size_t)(k - beg) * 9 + q] = (double)term; });
                for (const PixelFlip& f : frag) {
                    R Tf; uint32_t lastf;
                    pixel_forward<R>(g, beg, end, pixfx, pixfy, f, Tf, lastf, nullptr, 0.0f);
                    t1.assign(n, 0.0);
                    pixel_backward<R>(g, beg, end, pixfx, pixfy, Tf, lastf, bg, dpx, ddelx_dx, ddely_dy, f,
                                      [&](uint32_t k, int q, R term) { t1[(size_t)(k - beg) * 9 + q] = (double)term; });
                    for (size_t j = 0; j < n; j++) partflip[(size_t)beg * 9 + j] += std::fabs(t1[j] - t0[j]);
                    // the flipped pixel's terms are fp32 sums with a conditioning of their own (a splat blended only THROUGH a flip has
                    // nothing in the nominal bound): it joins cond9 for every term the flip moved
                    if (cond9)
                        pixel_cond<R>(g, beg, end, pixfx, pixfy, Tf, lastf, bg, dpx, ddelx_dx, ddely_dy, f, [&](uint32_t k, int q, double c) {
                            const size_t j = (size_t)(k - beg) * 9 + q;
                            if (t1[j] != t0[j]) partcond[(size_t)k * 9 + q] += c;
                        });
                }
            }
    }
    std::vector<double> acc((size_t)P * 9, 0.0);
    if (abs9) std::fill(abs9, abs9 + (size_t)P * 9, 0.0);
    if (flip9) std::fill(flip9, flip9 + (size_t)P * 9, 0.0);
    if (cond9) std::fill(cond9, cond9 + (size_t)P * 9, 0.0);
    for (size_t k = 0; k < Rn; k++) {
        const uint32_t id = g.point_list[k];
        for (int q = 0; q < 9; q++) acc[(size_t)id * 9 + q] += part[k * 9 + q];
        if (abs9) for (int q = 0; q < 9; q++) abs9[(size_t)id * 9 + q] += partabs[k * 9 + q];
        if (flip9) for (int q = 0; q < 9; q++) flip9[(size_t)id * 9 + q] += partflip[k * 9 + q];
        if (cond9) for (int q = 0; q < 9; q++) cond9[(size_t)id * 9 + q] += partcond[k * 9 + q];
    }
    for (int i = 0; i < P; i++) {
        const double* a = &acc[(size_t)i * 9];
        o.dL_dcolor[3 * (size_t)i] += (R)a[0]; o.dL_dcolor[3 * (size_t)i + 1] += (R)a[1]; o.dL_dcolor[3 * (size_t)i + 2] += (R)a[2];
        o.dL_dmean2D[3 * (size_t)i] += (R)a[3]; o.dL_dmean2D[3 * (size_t)i + 1] += (R)a[4];
        o.dL_dconic[4 * (size_t)i] += (R)a[5]; o.dL_dconic[4 * (size_t)i + 1] += (R)a[6]; o.dL_dconic[4 * (size_t)i + 3] += (R)a[7];
        o.dL_dopacity[i] += (R)a[8];
    }
}

// ---------------------------------------------------------------------------------------------
// The reference's OWN summation (test-side service, not part of the restatement above).  Upstream's backward render
// kernel — the call at src/Trainer.cu:378-412 — has one thread per pixel walk its tile's list back to front and add the
// nine terms of every blended (pixel, splat) pair to the splat's global accumulators with fp32 atomicAdd (buffers
// pre-zeroed at src/Trainer.cu:366-375).  The terms are deterministic fp32 values (pixel_backward<float> below is that
// arithmetic); the ORDER in which a splat's accumulator receives them is whatever the hardware retires first, so two runs of the
// reference differ by the rounding of two different fp32 summation orders.  render_backward above avoids the question by
// summing in double; this block answers it: atomic_prepare keeps every emitted term, atomic_sums adds a splat's terms in
// fp32 in a seeded order.  K seeds = K admissible runs of the reference; their per-entry [min, max] is the reference's
// run-to-run envelope, the yardstick the parity tolerances are calibrated against (tests/test_reference_noise.py).
//   mode 0: every (tile, pixel) term of a splat in one uniformly random order (tiles run concurrently on the device);
//   mode 1: tiles in random order, the pixels of one tile in random order, tile after tile (a device that runs few tiles at once);
//   mode 2: ascending tile, raster pixel order (one fixed admissible order);   mode 3: the reverse of mode 2.
// ---------------------------------------------------------------------------------------------
struct AtomicRec { float t[9]; };
struct AtomicTerms {
    std::vector<AtomicRec> rec;          // grouped by list entry k (= one (tile, splat) pair), pixel order inside
    std::vector<uint64_t> entry_off;     // [Rn + 1]
    std::vector<uint32_t> splat_entry;   // list positions k of every splat, ascending (= ascending tile)
    std::vector<uint64_t> splat_off;     // [P + 1]
    int P = 0;
};

inline uint64_t splitmix64(uint64_t& x) {
    uint64_t z = (x += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
template <class It> inline void seeded_shuffle(It first, size_t n, uint64_t& st) {
    for (size_t i = n; i > 1; i--) {
        const size_t j = (size_t)(splitmix64(st) % (uint64_t)i);   // (modulo bias ~ n / 2^64: none that matters)
        std::swap(first[i - 1], first[j]);
    }
}

uint64_t atomic_prepare(const ViewState<float>& g, const float* bg, const float* dL_dpix, AtomicTerms& at) {
    const int W = g.W, H = g.H, P = g.P;
    const size_t N = (size_t)W * H;
    const size_t Rn = g.point_list.size();
    const int T = g.gx * g.gy;
    const float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;
    struct TileRec { uint32_t k; AtomicRec r; };
    std::vector<std::vector<TileRec>> per_tile(T);
    std::vector<uint64_t> count(Rn + 1, 0);
#pragma omp parallel for schedule(dynamic, 4)
    for (int tile = 0; tile < T; tile++) {
        const int tx = tile % g.gx, ty = tile / g.gx;
        const uint32_t beg = g.ranges[2 * (size_t)tile], end = g.ranges[2 * (size_t)tile + 1];
        if (end <= beg) continue;
        std::vector<TileRec>& out = per_tile[tile];
        for (int py = ty * TILE; py < std::min(H, (ty + 1) * TILE); py++)
            for (int px = tx * TILE; px < std::min(W, (tx + 1) * TILE); px++) {
                const size_t pix = (size_t)py * W + px;
                const float dpx[3] = { dL_dpix[pix], dL_dpix[N + pix], dL_dpix[2 * N + pix] };
                pixel_backward<float>(g, beg, end, (float)px, (float)py, g.final_T[pix], g.n_contrib[pix], bg, dpx, ddelx_dx, ddely_dy, PixelFlip{},
                                      [&](uint32_t k, int q, float term) {
                                          if (q == 0) { out.push_back(TileRec{}); out.back().k = k; }
                                          out.back().r.t[q] = term;
                                      });
            }
        for (const TileRec& tr : out) count[tr.k]++;     // k in [beg, end): this tile's own counters
    }
    at.entry_off.assign(Rn + 1, 0);
    for (size_t k = 0; k < Rn; k++) at.entry_off[k + 1] = at.entry_off[k] + count[k];
    at.rec.resize(at.entry_off[Rn]);
#pragma omp parallel for schedule(dynamic, 4)
    for (int tile = 0; tile < T; tile++) {
        std::vector<TileRec>& out = per_tile[tile];
        if (out.empty()) continue;
        const uint32_t beg = g.ranges[2 * (size_t)tile], end = g.ranges[2 * (size_t)tile + 1];
        std::vector<uint64_t> cur(at.entry_off.begin() + beg, at.entry_off.begin() + end);
        for (const TileRec& tr : out) at.rec[cur[tr.k - beg]++] = tr.r;
        std::vector<TileRec>().swap(out);
    }
    at.P = P;
    at.splat_off.assign((size_t)P + 1, 0);
    for (size_t k = 0; k < Rn; k++) at.splat_off[(size_t)g.point_list[k] + 1]++;
    for (int i = 0; i < P; i++) at.splat_off[(size_t)i + 1] += at.splat_off[i];
    at.splat_entry.resize(Rn);
    std::vector<uint64_t> cur(at.splat_off.begin(), at.splat_off.end() - 1);
    for (size_t k = 0; k < Rn; k++) at.splat_entry[cur[g.point_list[k]]++] = (uint32_t)k;
    return (uint64_t)at.rec.size();
}

void atomic_sums(const AtomicTerms& at, uint64_t seed, int mode, float* sums9) {
    const int P = at.P;
#pragma omp parallel for schedule(dynamic, 64)
    for (int i = 0; i < P; i++) {
        float acc[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };    // the pre-zeroed accumulators, src/Trainer.cu:366-375
        const uint64_t e0 = at.splat_off[i], e1 = at.splat_off[(size_t)i + 1];
        uint64_t st = seed * 0xD1342543DE82EF95ull + (uint64_t)i * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
        std::vector<uint64_t> order;
        std::vector<uint32_t> ents(at.splat_entry.begin() + e0, at.splat_entry.begin() + e1);
        if (mode == 1) seeded_shuffle(ents.begin(), ents.size(), st);
        for (uint32_t k : ents) {
            const size_t first = order.size();
            for (uint64_t r = at.entry_off[k]; r < at.entry_off[(size_t)k + 1]; r++) order.push_back(r);
            if (mode == 1) seeded_shuffle(order.begin() + first, order.size() - first, st);
        }
        if (mode == 0) seeded_shuffle(order.begin(), order.size(), st);
        if (mode == 3) std::reverse(order.begin(), order.end());
        for (uint64_t r : order) {
            const float* t = at.rec[r].t;
            for (int q = 0; q < 9; q++) acc[q] = acc[q] + t[q];     // one fp32 round-to-nearest add per atomicAdd
        }
        for (int q = 0; q < 9; q++) sums9[(size_t)i * 9 + q] = acc[q];
    }
}

// A.8 + A.9 — per-splat backward: conic -> cov2D -> cov3D & mean; mean2D -> mean; colour -> SH &
// mean; cov3D -> scale & rotation (gradient w.r.t. the unnormalised quaternion).
template <class R>
void preprocess_backward(const ViewState<R>& g, int D, int M, const R* means, const R* scales, R mod, const R* rots,
                         const R* shs, const R* view, const R* proj, const R* campos, R tanx, R tany,
                         const Grads<R>& o) {
    const int P = g.P, W = g.W, H = g.H;
    const R focal_x = R(W) / (R(2.0) * tanx);
    const R focal_y = R(H) / (R(2.0) * tany);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < P; i++) {
        if (!(g.radii[i] > 0)) continue;
        const V3<R> mean = { means[3 * i], means[3 * i + 1], means[3 * i + 2] };
        // ---- cov2D backward (kernel 1 upstream) ----
        const R* c3 = &g.cov3D[6 * (size_t)i];
        const R gcx = o.dL_dconic[4 * (size_t)i], gcy = o.dL_dconic[4 * (size_t)i + 1], gcz = o.dL_dconic[4 * (size_t)i + 3];
        V3<R> t = transformPoint4x3(mean, view);
        const R limx = R(1.3) * tanx, limy = R(1.3) * tany;
        const R txtz = t.x / t.z, tytz = t.y / t.z;
        t.x = std::min(limx, std::max(-limx, txtz)) * t.z;
        t.y = std::min(limy, std::max(-limy, tytz)) * t.z;
        const R x_grad_mul = (txtz < -limx || txtz > limx) ? R(0) : R(1);
        const R y_grad_mul = (tytz < -limy || tytz > limy) ? R(0) : R(1);
        const R J00 = focal_x / t.z, J02 = -(focal_x * t.x) / (t.z * t.z);
        const R J11 = focal_y / t.z, J12 = -(focal_y * t.y) / (t.z * t.z);
        R T[2][3];
        for (int rr = 0; rr < 3; rr++) {
            T[0][rr] = view[4 * rr] * J00 + view[4 * rr + 2] * J02;
            T[1][rr] = view[4 * rr + 1] * J11 + view[4 * rr + 2] * J12;
        }
        const R V[3][3] = { { c3[0], c3[1], c3[2] }, { c3[1], c3[3], c3[4] }, { c3[2], c3[4], c3[5] } };
        R A[3][2];
        for (int k = 0; k < 3; k++)
            for (int rr = 0; rr < 2; rr++) A[k][rr] = T[rr][0] * V[0][k] + T[rr][1] * V[1][k] + T[rr][2] * V[2][k];
        const R a = A[0][0] * T[0][0] + A[1][0] * T[0][1] + A[2][0] * T[0][2] + R(0.3);
        const R b = A[0][1] * T[0][0] + A[1][1] * T[0][1] + A[2][1] * T[0][2];
        const R c = A[0][1] * T[1][0] + A[1][1] * T[1][1] + A[2][1] * T[1][2] + R(0.3);
        const R denom = a * c - b * b;
        R dL_da = 0, dL_db = 0, dL_dc = 0;
        const R denom2inv = R(1.0) / ((denom * denom) + R(0.0000001));
        R* dcov = &o.dL_dcov3D[6 * (size_t)i];
        if (denom2inv != R(0)) {
            dL_da = denom2inv * (-c * c * gcx + R(2) * b * c * gcy + (denom - a * c) * gcz);
            dL_dc = denom2inv * (-a * a * gcz + R(2) * a * b * gcy + (denom - a * c) * gcx);
            dL_db = denom2inv * R(2) * (b * c * gcx - (denom + R(2) * b * b) * gcy + a * b * gcz);
            dcov[0] = T[0][0] * T[0][0] * dL_da + T[0][0] * T[1][0] * dL_db + T[1][0] * T[1][0] * dL_dc;
            dcov[3] = T[0][1] * T[0][1] * dL_da + T[0][1] * T[1][1] * dL_db + T[1][1] * T[1][1] * dL_dc;
            dcov[5] = T[0][2] * T[0][2] * dL_da + T[0][2] * T[1][2] * dL_db + T[1][2] * T[1][2] * dL_dc;
            dcov[1] = R(2) * T[0][0] * T[0][1] * dL_da + (T[0][0] * T[1][1] + T[0][1] * T[1][0]) * dL_db + R(2) * T[1][0] * T[1][1] * dL_dc;
            dcov[2] = R(2) * T[0][0] * T[0][2] * dL_da + (T[0][0] * T[1][2] + T[0][2] * T[1][0]) * dL_db + R(2) * T[1][0] * T[1][2] * dL_dc;
            dcov[4] = R(2) * T[0][2] * T[0][1] * dL_da + (T[0][1] * T[1][2] + T[0][2] * T[1][1]) * dL_db + R(2) * T[1][1] * T[1][2] * dL_dc;
        } else {
            for (int k = 0; k < 6; k++) dcov[k] = 0;
        }
        // dL/dT (rows 0,1), then dL/dJ, then dL/dt, then dL/dmean (assignment, not accumulation)
        R dT[2][3];
        for (int k = 0; k < 3; k++) {
            const R tv0 = T[0][0] * V[k][0] + T[0][1] * V[k][1] + T[0][2] * V[k][2];
            const R tv1 = T[1][0] * V[k][0] + T[1][1] * V[k][1] + T[1][2] * V[k][2];
            dT[0][k] = R(2) * tv0 * dL_da + tv1 * dL_db;
            dT[1][k] = R(2) * tv1 * dL_dc + tv0 * dL_db;
        }
        // W[k][r] = view[4r+k]; dL_dJ(c,k) = sum_r W[k][r] * dT[c][r]
        const R dJ00 = view[0] * dT[0][0] + view[4] * dT[0][1] + view[8] * dT[0][2];
        const R dJ02 = view[2] * dT[0][0] + view[6] * dT[0][1] + view[10] * dT[0][2];
        const R dJ11 = view[1] * dT[1][0] + view[5] * dT[1][1] + view[9] * dT[1][2];
        const R dJ12 = view[2] * dT[1][0] + view[6] * dT[1][1] + view[10] * dT[1][2];
        const R tz = R(1) / t.z, tz2 = tz * tz, tz3 = tz2 * tz;
        const R dtx = x_grad_mul * -focal_x * tz2 * dJ02;
        const R dty = y_grad_mul * -focal_y * tz2 * dJ12;
        const R dtz = -focal_x * tz2 * dJ00 - focal_y * tz2 * dJ11 + (R(2) * focal_x * t.x) * tz3 * dJ02 +
                      (R(2) * focal_y * t.y) * tz3 * dJ12;
        V3<R> dmean = transformVec4x3Transpose(V3<R>{ dtx, dty, dtz }, view);

        // ---- kernel 2 upstream: projection of mean2D gradient ----
        const V4<R> mh = transformPoint4x4(mean, proj);
        const R m_w = R(1.0) / (mh.w + R(0.0000001));
        const R mul1 = (proj[0] * mean.x + proj[4] * mean.y + proj[8] * mean.z + proj[12]) * m_w * m_w;
        const R mul2 = (proj[1] * mean.x + proj[5] * mean.y + proj[9] * mean.z + proj[13]) * m_w * m_w;
        const R g2x = o.dL_dmean2D[3 * (size_t)i], g2y = o.dL_dmean2D[3 * (size_t)i + 1];
        dmean.x += (proj[0] * m_w - proj[3] * mul1) * g2x + (proj[1] * m_w - proj[3] * mul2) * g2y;
        dmean.y += (proj[4] * m_w - proj[7] * mul1) * g2x + (proj[5] * m_w - proj[7] * mul2) * g2y;
        dmean.z += (proj[8] * m_w - proj[11] * mul1) * g2x + (proj[9] * m_w - proj[11] * mul2) * g2y;

        // ---- SH backward ----
        {
            const V3<R> dir_orig = { mean.x - campos[0], mean.y - campos[1], mean.z - campos[2] };
            const R len = std::sqrt(dir_orig.x * dir_orig.x + dir_orig.y * dir_orig.y + dir_orig.z * dir_orig.z);
            const R X = dir_orig.x / len, Y = dir_orig.y / len, Z = dir_orig.z / len;
            const R* sh = shs + (size_t)i * M * 3;
            R* dsh = o.dL_dsh + (size_t)i * M * 3;
            R dRGB[3];
            for (int ch = 0; ch < 3; ch++) dRGB[ch] = o.dL_dcolor[3 * (size_t)i + ch] * (g.clamped[3 * (size_t)i + ch] ? R(0) : R(1));
            R ddir[3] = { 0, 0, 0 };
            for (int ch = 0; ch < 3; ch++) {
                const R dl = dRGB[ch];
                R dx_ = 0, dy_ = 0, dz_ = 0;
                dsh[ch] = SH<R>::C0 * dl;
                if (D > 0) {
                    dsh[3 + ch] = (-SH<R>::C1 * Y) * dl; dsh[6 + ch] = (SH<R>::C1 * Z) * dl; dsh[9 + ch] = (-SH<R>::C1 * X) * dl;
                    dx_ = -SH<R>::C1 * sh[9 + ch]; dy_ = -SH<R>::C1 * sh[3 + ch]; dz_ = SH<R>::C1 * sh[6 + ch];
                    if (D > 1) {
                        const R xx = X * X, yy = Y * Y, zz = Z * Z, xy = X * Y, yz = Y * Z, xz = X * Z;
                        dsh[12 + ch] = (SH<R>::C2[0] * xy) * dl; dsh[15 + ch] = (SH<R>::C2[1] * yz) * dl;
                        dsh[18 + ch] = (SH<R>::C2[2] * (R(2) * zz - xx - yy)) * dl; dsh[21 + ch] = (SH<R>::C2[3] * xz) * dl;
                        dsh[24 + ch] = (SH<R>::C2[4] * (xx - yy)) * dl;
                        dx_ += SH<R>::C2[0] * Y * sh[12 + ch] + SH<R>::C2[2] * R(2) * -X * sh[18 + ch] + SH<R>::C2[3] * Z * sh[21 + ch] + SH<R>::C2[4] * R(2) * X * sh[24 + ch];
                        dy_ += SH<R>::C2[0] * X * sh[12 + ch] + SH<R>::C2[1] * Z * sh[15 + ch] + SH<R>::C2[2] * R(2) * -Y * sh[18 + ch] + SH<R>::C2[4] * R(2) * -Y * sh[24 + ch];
                        dz_ += SH<R>::C2[1] * Y * sh[15 + ch] + SH<R>::C2[2] * R(2) * R(2) * Z * sh[18 + ch] + SH<R>::C2[3] * X * sh[21 + ch];
                        if (D > 2) {
                            dsh[27 + ch] = (SH<R>::C3[0] * Y * (R(3) * xx - yy)) * dl;
                            dsh[30 + ch] = (SH<R>::C3[1] * xy * Z) * dl;
                            dsh[33 + ch] = (SH<R>::C3[2] * Y * (R(4) * zz - xx - yy)) * dl;
                            dsh[36 + ch] = (SH<R>::C3[3] * Z * (R(2) * zz - R(3) * xx - R(3) * yy)) * dl;
                            dsh[39 + ch] = (SH<R>::C3[4] * X * (R(4) * zz - xx - yy)) * dl;
                            dsh[42 + ch] = (SH<R>::C3[5] * Z * (xx - yy)) * dl;
                            dsh[45 + ch] = (SH<R>::C3[6] * X * (xx - R(3) * yy)) * dl;
                            dx_ += SH<R>::C3[0] * sh[27 + ch] * R(3) * R(2) * xy + SH<R>::C3[1] * sh[30 + ch] * yz +
                                   SH<R>::C3[2] * sh[33 + ch] * -R(2) * xy + SH<R>::C3[3] * sh[36 + ch] * -R(3) * R(2) * xz +
                                   SH<R>::C3[4] * sh[39 + ch] * (-R(3) * xx + R(4) * zz - yy) + SH<R>::C3[5] * sh[42 + ch] * R(2) * xz +
                                   SH<R>::C3[6] * sh[45 + ch] * R(3) * (xx - yy);
                            dy_ += SH<R>::C3[0] * sh[27 + ch] * R(3) * (xx - yy) + SH<R>::C3[1] * sh[30 + ch] * xz +
                                   SH<R>::C3[2] * sh[33 + ch] * (-R(3) * yy + R(4) * zz - xx) + SH<R>::C3[3] * sh[36 + ch] * -R(3) * R(2) * yz +
                                   SH<R>::C3[4] * sh[39 + ch] * -R(2) * xy + SH<R>::C3[5] * sh[42 + ch] * -R(2) * yz +
                                   SH<R>::C3[6] * sh[45 + ch] * -R(3) * R(2) * xy;
                            dz_ += SH<R>::C3[1] * sh[30 + ch] * xy + SH<R>::C3[2] * sh[33 + ch] * R(4) * R(2) * yz +
                                   SH<R>::C3[3] * sh[36 + ch] * R(3) * (R(2) * zz - xx - yy) + SH<R>::C3[4] * sh[39 + ch] * R(4) * R(2) * xz +
                                   SH<R>::C3[5] * sh[42 + ch] * (xx - yy);
                        }
                    }
                }
                ddir[0] += dx_ * dl; ddir[1] += dy_ * dl; ddir[2] += dz_ * dl;
            }
            // through the normalisation (dnormvdv upstream)
            const R sum2 = dir_orig.x * dir_orig.x + dir_orig.y * dir_orig.y + dir_orig.z * dir_orig.z;
            const R invsum32 = R(1.0) / std::sqrt(sum2 * sum2 * sum2);
            const V3<R>& v = dir_orig;
            dmean.x += ((+sum2 - v.x * v.x) * ddir[0] - v.y * v.x * ddir[1] - v.z * v.x * ddir[2]) * invsum32;
            dmean.y += (-v.x * v.y * ddir[0] + (sum2 - v.y * v.y) * ddir[1] - v.z * v.y * ddir[2]) * invsum32;
            dmean.z += (-v.x * v.z * ddir[0] - v.y * v.z * ddir[1] + (sum2 - v.z * v.z) * ddir[2]) * invsum32;
        }
        o.dL_dmean3D[3 * (size_t)i] = dmean.x; o.dL_dmean3D[3 * (size_t)i + 1] = dmean.y; o.dL_dmean3D[3 * (size_t)i + 2] = dmean.z;

        // ---- cov3D backward: Sigma = M^T M, M = S Rg ----
        {
            const R sx = mod * scales[3 * i], sy = mod * scales[3 * i + 1], sz = mod * scales[3 * i + 2];
            const R s[3] = { sx, sy, sz };
            const R r = rots[4 * i], x = rots[4 * i + 1], y = rots[4 * i + 2], z = rots[4 * i + 3];
            R Rg[3][3];
            Rg[0][0] = R(1.0) - R(2.0) * (y * y + z * z); Rg[0][1] = R(2.0) * (x * y - r * z); Rg[0][2] = R(2.0) * (x * z + r * y);
            Rg[1][0] = R(2.0) * (x * y + r * z); Rg[1][1] = R(1.0) - R(2.0) * (x * x + z * z); Rg[1][2] = R(2.0) * (y * z - r * x);
            Rg[2][0] = R(2.0) * (x * z - r * y); Rg[2][1] = R(2.0) * (y * z + r * x); Rg[2][2] = R(1.0) - R(2.0) * (x * x + y * y);
            R Mm[3][3];
            for (int cc = 0; cc < 3; cc++) for (int k = 0; k < 3; k++) Mm[cc][k] = s[k] * Rg[cc][k];
            // symmetric dL/dSigma with halved off-diagonals
            const R dS[3][3] = { { dcov[0], R(0.5) * dcov[1], R(0.5) * dcov[2] },
                                 { R(0.5) * dcov[1], dcov[3], R(0.5) * dcov[4] },
                                 { R(0.5) * dcov[2], R(0.5) * dcov[4], dcov[5] } };
            // dL_dM = 2 * M * dSigma (glm): dM[c][k] = 2 * sum_j M[j][k] * dS[c][j]
            R dM[3][3];
            for (int cc = 0; cc < 3; cc++)
                for (int k = 0; k < 3; k++) dM[cc][k] = R(2.0) * (Mm[0][k] * dS[cc][0] + Mm[1][k] * dS[cc][1] + Mm[2][k] * dS[cc][2]);
            // Rt[k][c] = Rg[c][k], dMt[k][c] = dM[c][k]; dL_dscale_k = dot(Rt[k], dMt[k])
            R dMt[3][3];
            for (int k = 0; k < 3; k++) for (int cc = 0; cc < 3; cc++) dMt[k][cc] = dM[cc][k];
            for (int k = 0; k < 3; k++)
                o.dL_dscale[3 * (size_t)i + k] = Rg[0][k] * dMt[k][0] + Rg[1][k] * dMt[k][1] + Rg[2][k] * dMt[k][2];
            for (int k = 0; k < 3; k++) for (int cc = 0; cc < 3; cc++) dMt[k][cc] *= s[k];
            R* dq = &o.dL_drot[4 * (size_t)i];
            dq[0] = R(2) * z * (dMt[0][1] - dMt[1][0]) + R(2) * y * (dMt[2][0] - dMt[0][2]) + R(2) * x * (dMt[1][2] - dMt[2][1]);
            dq[1] = R(2) * y * (dMt[1][0] + dMt[0][1]) + R(2) * z * (dMt[2][0] + dMt[0][2]) + R(2) * r * (dMt[1][2] - dMt[2][1]) - R(4) * x * (dMt[2][2] + dMt[1][1]);
            dq[2] = R(2) * x * (dMt[1][0] + dMt[0][1]) + R(2) * r * (dMt[2][0] - dMt[0][2]) + R(2) * z * (dMt[1][2] + dMt[2][1]) - R(4) * y * (dMt[2][2] + dMt[0][0]);
            dq[3] = R(2) * r * (dMt[0][1] - dMt[1][0]) + R(2) * x * (dMt[2][0] + dMt[0][2]) + R(2) * y * (dMt[1][2] + dMt[2][1]) - R(4) * z * (dMt[1][1] + dMt[0][0]);
        }
    }
}

// Test-side service: the ADMISSIBLE forward results of one pixel.  The blend takes two discrete decisions per (pixel, entry)
// pair — "alpha < 1/255: skip" and "T (1 - alpha) < 1e-4: stop, entry not applied" — and an implementation whose exp() or
// running product differs from this one's in the last bits may take the other branch where a pair sits on a threshold.
// pixel_run blends the pixel with a list of FORCED decisions; at the first decision that is fragile (within alpha_margin /
// T_margin, relative, of its threshold) and not in the list it stops and reports it.  Exploring both values of every
// reported decision depth-first enumerates the admissible blends of the pixel (orc_check_pixels_f32): a foreign result
// has to equal ONE of them — colour, final T and last contributor — instead of being excluded from the comparison.
struct ForcedDecision { uint32_t k; int kind; bool value; };  // kind 1: skip (alpha test), 2: stop (T test), 3: skip (power > 0; only with exp_cond)
template <class R> struct PixelLeaf { R T, C[3]; uint32_t last; R tolT, tolC[3]; };
// returns true when the run completed (leaf filled); false when it stopped at an unforced fragile decision (*branch, with its
// nominal value)
template <class R>
inline bool pixel_run(const ViewState<R>& g, uint32_t beg, uint32_t end, R pixfx, R pixfy, const std::vector<ForcedDecision>& forced,
                      float alpha_margin, float T_margin, PixelLeaf<R>& leaf, ForcedDecision* branch, float exp_cond = 0.0f) {
    // exp_cond > 0: the conditioning of the exponent.  power = -(a dx^2 + c dy^2)/2 - b dx dy is a sum of three products that
    // cancel for a splat seen far along its long axis (|a| dx^2 / 2 + |c| dy^2 / 2 + |b dx dy| = m >> |power|): every fp32 evaluation
    // order, the reference's included, carries ~2^-24 m of absolute error in the power, i.e. a RELATIVE error of that size in alpha.
    // An implementation may deviate by exp_cond x 2^-24 m_i in alpha_i (none where alpha is clipped to 0.99); carried to first
    // order through the blend that is tolC / tolT of the leaf:  dC = sum_i rho_i alpha_i |c_i T_i - S_i / (1 - alpha_i)|
    // (S_i: what lies behind entry i),  dT = T sum_i rho_i alpha_i / (1 - alpha_i).
    struct Term { R alpha, T, c[3], rho; };
    std::vector<Term> terms;
    R Tt = R(1.0), C[3] = { 0, 0, 0 };
    uint32_t contributor = 0, last = 0;
    double tcond = 0.0;
    auto forced_value = [&](uint32_t k, int kind, bool& v) {
        for (const ForcedDecision& f : forced) if (f.k == k && f.kind == kind) { v = f.value; return true; }
        return false;
    };
    for (uint32_t k = beg; k < end; k++) {
        contributor++;
        const uint32_t id = g.point_list[k];
        const R dx = g.means2D[2 * (size_t)id] - pixfx, dy = g.means2D[2 * (size_t)id + 1] - pixfy;
        const R* co = &g.conic_opacity[4 * (size_t)id];
        const R power = R(-0.5) * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
        bool pskip = power > R(0.0);
        if (!forced_value(k, 3, pskip) && exp_cond > 0.0f && power_ulps_from_zero<R>(co, dx, dy, power) < exp_cond) {
            *branch = { k, 3, pskip };
            return false;
        }
        if (pskip) continue;
        const R alpha = std::min(R(0.99), co[3] * std::exp(power));
        const double mk = (exp_cond > 0.0f && alpha < R(0.99)) ? power_magnitude<R>(co, dx, dy) : 0.0;
        bool skip = alpha < R(1.0) / R(255.0);
        if (!forced_value(k, 1, skip) && (double)(std::fabs(alpha - R(1.0 / 255.0)) * R(255.0)) < (double)alpha_margin + exp_cond * ULP24 * mk) {
            *branch = { k, 1, skip };
            return false;
        }
        if (skip) continue;
        const R test_T = Tt * (R(1.0) - alpha);
        bool stop = test_T < R(0.0001);
        tcond += mk * (double)alpha / (1.0 - (double)alpha) + 1.0;
        if (!forced_value(k, 2, stop) && (double)(std::fabs(test_T - R(0.0001)) * R(10000.0)) < (double)T_margin + exp_cond * ULP24 * tcond) {
            *branch = { k, 2, stop };
            return false;
        }
        if (stop) break;
        for (int c = 0; c < 3; c++) C[c] += g.rgb[3 * (size_t)id + c] * alpha * Tt;
        if (exp_cond > 0.0f) {
            const R m = R(0.5) * (std::fabs(co[0]) * dx * dx + std::fabs(co[2]) * dy * dy) + std::fabs(co[1] * dx * dy);
            terms.push_back({ alpha, Tt, { g.rgb[3 * (size_t)id], g.rgb[3 * (size_t)id + 1], g.rgb[3 * (size_t)id + 2] },
                              alpha < R(0.99) ? R(exp_cond) * R(5.9604644775390625e-08) * m : R(0) });
        }
        Tt = test_T;
        last = contributor;
    }
    leaf.T = Tt; leaf.last = last;
    leaf.tolT = 0;
    for (int c = 0; c < 3; c++) { leaf.C[c] = C[c]; leaf.tolC[c] = 0; }
    R behind[3] = { 0, 0, 0 };
    for (size_t i = terms.size(); i-- > 0;) {
        const Term& t = terms[i];
        leaf.tolT += t.rho * t.alpha / (R(1.0) - t.alpha);
        for (int c = 0; c < 3; c++) {
            leaf.tolC[c] += t.rho * t.alpha * std::fabs(t.c[c] * t.T - behind[c] / (R(1.0) - t.alpha));
            behind[c] += t.c[c] * t.alpha * t.T;
        }
    }
    leaf.tolT *= Tt;
    // one ulp per factor of the running product T (and of the weights alpha T the colour is made of)
    const R per_factor = R(exp_cond) * R(5.9604644775390625e-08) * R(terms.size());
    leaf.tolT += per_factor * Tt;
    for (int c = 0; c < 3; c++) leaf.tolC[c] += per_factor * C[c];
    return true;
}

// Every pixel against its admissible blends: matches(pix, leaf) says whether a foreign result equals the blend `leaf`.  status[pix]: 0 the
// nominal blend matched, 1 another admissible blend did, 2 none, 3 undecided (more than max_leaves blends); returns the count of status >= 2.
template <class Match>
int check_admissible(const ViewState<float>& g, float alpha_margin, float T_margin, int max_leaves, float exp_cond, int32_t* status, int32_t* leaves,
                     Match matches) {
    const int W = g.W, H = g.H;
    const int T = g.gx * g.gy;
    int n_bad = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : n_bad)
    for (int tile = 0; tile < T; tile++) {
        const int tx = tile % g.gx, ty = tile / g.gx;
        const uint32_t beg = g.ranges[2 * (size_t)tile], end = g.ranges[2 * (size_t)tile + 1];
        std::vector<std::vector<ForcedDecision>> stack;
        for (int py = ty * TILE; py < std::min(H, (ty + 1) * TILE); py++)
            for (int px = tx * TILE; px < std::min(W, (tx + 1) * TILE); px++) {
                const size_t pix = (size_t)py * W + px;
                stack.clear();
                stack.push_back({});
                int n_leaves = 0, st = 2;
                bool first = true;
                while (!stack.empty()) {
                    std::vector<ForcedDecision> forced = std::move(stack.back());
                    stack.pop_back();
                    PixelLeaf<float> leaf;
                    ForcedDecision br;
                    if (pixel_run<float>(g, beg, end, (float)px, (float)py, forced, alpha_margin, T_margin, leaf, &br, exp_cond)) {
                        n_leaves++;
                        // the nominal blend is the leaf reached by taking the nominal value at every branch: explored first
                        if (matches(pix, leaf)) { st = first ? 0 : 1; break; }
                        first = false;
                        if (n_leaves >= max_leaves) { st = 3; break; }
                    } else {
                        std::vector<ForcedDecision> other = forced;
                        other.push_back({ br.k, br.kind, !br.value });
                        forced.push_back(br);
                        stack.push_back(std::move(other));   // explored after ...
                        stack.push_back(std::move(forced));  // ... the nominal value (LIFO)
                    }
                }
                status[pix] = st; leaves[pix] = n_leaves;
                if (st >= 2) n_bad++;
            }
    }
    return n_bad;
}

template <class R> struct State { ViewState<R> v; };

template <class R>
int forward_impl(State<R>* st, int P, int D, int M, const R* bg, int W, int H, const R* means, const R* shs,
                 const R* opac, const R* scales, R mod, const R* rots, const R* view, const R* proj, const R* campos,
                 R tanx, R tany, R* out_color) {
    preprocess<R>(P, D, M, means, scales, mod, rots, opac, shs, view, proj, campos, W, H, tanx, tany, st->v);
    bin_and_sort<R>(st->v);
    render_forward<R>(st->v, bg, out_color);
    return st->v.R_total;
}

template <class R>
void backward_impl(State<R>* st, int D, int M, const R* bg, const R* means, const R* shs, const R* scales, R mod,
                   const R* rots, const R* view, const R* proj, const R* campos, R tanx, R tany, const R* dL_dpix,
                   const Grads<R>& o, double* abs9, double* flip9 = nullptr, float flip_margin = 0.0f, double* cond9 = nullptr, float power_ulps = 0.0f) {
    render_backward<R>(st->v, bg, dL_dpix, o, abs9, flip9, flip_margin, cond9, power_ulps);
    preprocess_backward<R>(st->v, D, M, means, scales, mod, rots, shs, view, proj, campos, tanx, tany, o);
}

// ---------------------------------------------------------------------------------------------
// glm restatements used by Camera.cpp / densify (glm 0.9.9 semantics, right-handed, -1..1 depth)
// ---------------------------------------------------------------------------------------------
struct f3 { float x, y, z; };
inline f3 sub3(f3 a, f3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline f3 cross3(f3 a, f3 b) { return { a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y }; }
inline f3 norm3(f3 a) { const float inv = 1.0f / std::sqrt(dot3(a, a)); return { a.x * inv, a.y * inv, a.z * inv }; }

// glm::mat3_cast of quat (w,x,y,z), no normalisation; out[col][row]
inline void quat_to_mat3(float w, float x, float y, float z, float out[3][3]) {
    const float qxx = x * x, qyy = y * y, qzz = z * z, qxz = x * z, qxy = x * y, qyz = y * z, qwx = w * x, qwy = w * y, qwz = w * z;
    out[0][0] = 1.0f - 2.0f * (qyy + qzz); out[0][1] = 2.0f * (qxy + qwz); out[0][2] = 2.0f * (qxz - qwy);
    out[1][0] = 2.0f * (qxy - qwz); out[1][1] = 1.0f - 2.0f * (qxx + qzz); out[1][2] = 2.0f * (qyz + qwx);
    out[2][0] = 2.0f * (qxz + qwy); out[2][1] = 2.0f * (qyz - qwx); out[2][2] = 1.0f - 2.0f * (qxx + qyy);
}

}  // namespace

// =============================================================================================
// C interface (ctypes)
// =============================================================================================
extern "C" {

struct orc_state { State<float> f; State<double> d; AtomicTerms at; };

orc_state* orc_state_new() { return new orc_state(); }
void orc_state_free(orc_state* s) { delete s; }

int orc_num_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

// The thread count of every parallel region below.  pyoracle sets it to the CPUs the process may actually USE (affinity and
// cgroup quota): a GPU box of the pool shows 256 hardware threads to a container that is allowed 16 CPUs of run time, and
// OpenMP's default of 256 threads then spends its life in barriers (measured there: 13.7 s instead of 0.03 s for the budget of
// a 2500-splat scene, and 2.5-3 x on a full cfg3 pass).
void orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

// CudaRasterizer::Rasterizer::forward as called at src/Trainer.cu:334-360 (colors_precomp = cov3D_precomp =
// nullptr, prefiltered = false).  Returns num_rendered.
int orc_forward_f32(orc_state* s, int P, int D, int M, const float* bg, int W, int H, const float* means,
                    const float* shs, const float* opac, const float* scales, float mod, const float* rots,
                    const float* view, const float* proj, const float* campos, float tanx, float tany,
                    float* out_color) {
    return forward_impl<float>(&s->f, P, D, M, bg, W, H, means, shs, opac, scales, mod, rots, view, proj, campos, tanx, tany, out_color);
}
int orc_forward_f64(orc_state* s, int P, int D, int M, const double* bg, int W, int H, const double* means,
                    const double* shs, const double* opac, const double* scales, double mod, const double* rots,
                    const double* view, const double* proj, const double* campos, double tanx, double tany,
                    double* out_color) {
    return forward_impl<double>(&s->d, P, D, M, bg, W, H, means, shs, opac, scales, mod, rots, view, proj, campos, tanx, tany, out_color);
}

// CudaRasterizer::Rasterizer::backward as called at src/Trainer.cu:378-412.  Output buffers are
// accumulated into where the reference's are (caller pre-zeroes, src/Trainer.cu:366-375).
void orc_backward_f32(orc_state* s, int D, int M, const float* bg, const float* means, const float* shs,
                      const float* scales, float mod, const float* rots, const float* view, const float* proj,
                      const float* campos, float tanx, float tany, const float* dL_dpix, float* dL_dmean2D,
                      float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D,
                      float* dL_dsh, float* dL_dscale, float* dL_drot, double* abs9) {
    Grads<float> o{ dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot };
    backward_impl<float>(&s->f, D, M, bg, means, shs, scales, mod, rots, view, proj, campos, tanx, tany, dL_dpix, o, abs9);
}
// The same, plus the decision-flip allowance (see render_backward): flip9[P][9] receives, per splat and sum, the
// total |term change| over single-decision flips of every pair within `flip_margin` (relative) of a blend threshold.
void orc_backward_f32_flip(orc_state* s, int D, int M, const float* bg, const float* means, const float* shs,
                           const float* scales, float mod, const float* rots, const float* view, const float* proj,
                           const float* campos, float tanx, float tany, const float* dL_dpix, float* dL_dmean2D,
                           float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D,
                           float* dL_dsh, float* dL_dscale, float* dL_drot, double* abs9, double* flip9, float flip_margin) {
    Grads<float> o{ dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot };
    backward_impl<float>(&s->f, D, M, bg, means, shs, scales, mod, rots, view, proj, campos, tanx, tany, dL_dpix, o, abs9, flip9, flip_margin);
}
// The same, plus cond9[P][9]: the conditioning of every sum in units of 2^-24 (pixel_cond); power_ulps > 0 adds the pairs whose power
// lies within that many units of its own conditioning of zero to the decisions flip9 is taken over (PixelFlip kind 3).
void orc_backward_f32_cond(orc_state* s, int D, int M, const float* bg, const float* means, const float* shs,
                           const float* scales, float mod, const float* rots, const float* view, const float* proj,
                           const float* campos, float tanx, float tany, const float* dL_dpix, float* dL_dmean2D,
                           float* dL_dconic, float* dL_dopacity, float* dL_dcolor, float* dL_dmean3D, float* dL_dcov3D,
                           float* dL_dsh, float* dL_dscale, float* dL_drot, double* abs9, double* flip9, float flip_margin, double* cond9, float power_ulps) {
    Grads<float> o{ dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot };
    backward_impl<float>(&s->f, D, M, bg, means, shs, scales, mod, rots, view, proj, campos, tanx, tany, dL_dpix, o, abs9, flip9, flip_margin, cond9, power_ulps);
}
// The per-splat half of the backward alone (upstream's computeCov2DCUDA + preprocessCUDA backward)
// on caller-supplied pixel-stage sums: sums9[P][9] = dL_dcolor(3), dL_dmean2D(2), dL_dconic x/y/w(3), dL_dopacity(1).
// The chain is LINEAR in those sums for a fixed scene, so the parity tests can carry a per-splat tolerance on the
// sums (e.g. the decision-flip allowance) through to dL_dmean3D / dL_dcov3D / dL_dsh / dL_dscale / dL_drot by
// evaluating it on unit inputs.  Uses the state of the last orc_forward_f32.
void orc_chain_f32(orc_state* s, int D, int M, const float* means, const float* shs, const float* scales, float mod,
                   const float* rots, const float* view, const float* proj, const float* campos, float tanx, float tany,
                   const float* sums9, float* dL_dmean3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscale, float* dL_drot) {
    const int P = s->f.v.P;
    std::vector<float> m2((size_t)P * 3, 0.0f), con((size_t)P * 4, 0.0f), op(P, 0.0f), col((size_t)P * 3, 0.0f);
    for (int i = 0; i < P; i++) {
        const float* q = sums9 + (size_t)i * 9;
        col[3 * (size_t)i] = q[0]; col[3 * (size_t)i + 1] = q[1]; col[3 * (size_t)i + 2] = q[2];
        m2[3 * (size_t)i] = q[3]; m2[3 * (size_t)i + 1] = q[4];
        con[4 * (size_t)i] = q[5]; con[4 * (size_t)i + 1] = q[6]; con[4 * (size_t)i + 3] = q[7];
        op[i] = q[8];
    }
    Grads<float> o{ m2.data(), con.data(), op.data(), col.data(), dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot };
    preprocess_backward<float>(s->f.v, D, M, means, scales, mod, rots, shs, view, proj, campos, tanx, tany, o);
}
// The reference's own fp32 atomicAdd summation (see atomic_prepare above), on the state of the last orc_forward_f32:
// orc_atomic_prepare keeps every term upstream's backward render kernel would add for this dL_dpix and returns their number;
// orc_atomic_sums writes sums9[P][9] (layout of orc_chain_f32) summed in fp32 in the order (seed, mode) names;
// orc_atomic_release drops the terms.  The per-splat chain of such a run is orc_chain_f32 on those sums (it has no atomics upstream).
uint64_t orc_atomic_prepare(orc_state* s, const float* bg, const float* dL_dpix) { return atomic_prepare(s->f.v, bg, dL_dpix, s->at); }
void orc_atomic_sums(orc_state* s, uint64_t seed, int mode, float* sums9) { atomic_sums(s->at, seed, mode, sums9); }
void orc_atomic_release(orc_state* s) { s->at = AtomicTerms(); }
// Every pixel of a foreign implementation's forward output (colour [3][N] incl. background, final T, last contributor)
// against the admissible blends of that pixel (pixel_run above).  status[pix]:
//   0  equals the nominal blend (the one orc_forward_f32 produced)
//   1  equals another admissible blend: some fragile decision(s) taken the other way
//   2  equals NO admissible blend
//   3  undecided: more than max_leaves blends without a match
// leaves[pix] = blends examined (1: the pixel holds no fragile decision).  "Equals": last contributor identical,
// |T - T'| <= rtol max(|T'|, floor_T), |C - C'| <= rtol max(|C'|, floor_C) per channel — plus, with exp_cond > 0, the leaf's
// tolT / tolC (the conditioning of the exponent, see pixel_run).  Returns the number of pixels with status >= 2.
int orc_check_pixels_f32(orc_state* s, const float* bg, const float* got_color, const float* got_T, const uint32_t* got_last,
                         float alpha_margin, float T_margin, float rtol, float floor_T, float floor_C, int max_leaves,
                         int32_t* status, int32_t* leaves, float exp_cond) {
    const ViewState<float>& g = s->f.v;
    const size_t N = (size_t)g.W * g.H;
    return check_admissible(g, alpha_margin, T_margin, max_leaves, exp_cond, status, leaves, [&](size_t pix, const PixelLeaf<float>& l) {
        if (l.last != got_last[pix]) return false;
        if (!(std::fabs(got_T[pix] - l.T) <= rtol * std::max(std::fabs(l.T), floor_T) + l.tolT)) return false;
        for (int c = 0; c < 3; c++) {
            const float want = l.C[c] + l.T * bg[c];
            if (!(std::fabs(got_color[c * N + pix] - want) <= rtol * std::max(std::fabs(want), floor_C) + l.tolC[c] + l.tolT * std::fabs(bg[c]))) return false;
        }
        return true;
    });
}
// The same for an RGBA8 frame (imageFloatToInt, src/Trainer.cu:19-29: byte = clamp((int)(v * 256), 0, 255), R in the low byte) — what
// Trainer::render hands out: a pixel is accepted when, for SOME admissible blend of it, every channel's float — anywhere within the
// pixel tolerance (rtol max(|v|, floor_C) + the leaf's conditioning tolerance) of that blend's value — quantises to the byte the frame holds,
// i.e. the tolerance interval meets the byte's float interval [b / 256, (b + 1) / 256) (open-ended at 0 and 255).  Same status codes.
int orc_check_frame_f32(orc_state* s, const float* bg, const uint32_t* frame, float alpha_margin, float T_margin, float rtol, float floor_C,
                        int max_leaves, int32_t* status, int32_t* leaves, float exp_cond) {
    const ViewState<float>& g = s->f.v;
    return check_admissible(g, alpha_margin, T_margin, max_leaves, exp_cond, status, leaves, [&](size_t pix, const PixelLeaf<float>& l) {
        const uint32_t px = frame[pix];
        if ((px >> 24) != 0xFFu) return false;
        for (int c = 0; c < 3; c++) {
            const int b = (int)((px >> (8 * c)) & 0xFFu);
            const double want = (double)l.C[c] + (double)l.T * bg[c];
            const double tol = (double)rtol * std::max(std::fabs(want), (double)floor_C) + l.tolC[c] + l.tolT * std::fabs(bg[c]);
            const double lo = b == 0 ? -1e300 : b / 256.0, hi = b == 255 ? 1e300 : (b + 1) / 256.0;   // floats v with clamp((int)(v * 256)) == b
            if (!(want + tol >= lo && want - tol < hi)) return false;
        }
        return true;
    });
}
void orc_backward_f64(orc_state* s, int D, int M, const double* bg, const double* means, const double* shs,
                      const double* scales, double mod, const double* rots, const double* view, const double* proj,
                      const double* campos, double tanx, double tany, const double* dL_dpix, double* dL_dmean2D,
                      double* dL_dconic, double* dL_dopacity, double* dL_dcolor, double* dL_dmean3D, double* dL_dcov3D,
                      double* dL_dsh, double* dL_dscale, double* dL_drot) {
    Grads<double> o{ dL_dmean2D, dL_dconic, dL_dopacity, dL_dcolor, dL_dmean3D, dL_dcov3D, dL_dsh, dL_dscale, dL_drot };
    backward_impl<double>(&s->d, D, M, bg, means, shs, scales, mod, rots, view, proj, campos, tanx, tany, dL_dpix, o, nullptr);
}

// Copy a named piece of the fp32 per-view state out.  Returns bytes written, or -1 (unknown name /
// dst too small).
long orc_get(orc_state* s, const char* name, void* dst, long cap_bytes) {
    const ViewState<float>& g = s->f.v;
    const std::string n(name);
    const void* src = nullptr; size_t bytes = 0; bool found = false;
#define ORC_FIELD(field) if (n == #field) { found = true; src = g.field.data(); bytes = g.field.size() * sizeof(g.field[0]); }
    ORC_FIELD(depth) ORC_FIELD(means2D) ORC_FIELD(cov3D) ORC_FIELD(conic_opacity) ORC_FIELD(rgb) ORC_FIELD(radii)
    ORC_FIELD(tiles_touched) ORC_FIELD(point_offsets) ORC_FIELD(rect) ORC_FIELD(clamped) ORC_FIELD(keys_unsorted)
    ORC_FIELD(keys) ORC_FIELD(vals_unsorted) ORC_FIELD(point_list) ORC_FIELD(ranges) ORC_FIELD(final_T)
    ORC_FIELD(n_contrib) ORC_FIELD(margin) ORC_FIELD(splat_margin)
#undef ORC_FIELD
    if (!found || (long)bytes > cap_bytes) return -1;
    if (bytes) std::memcpy(dst, src, bytes);
    return (long)bytes;
}
int orc_num_rendered(orc_state* s) { return s->f.v.R_total; }

// ---------------------------------------------------------------------------------------------
// The reference's own four kernels (src/Trainer.cu:19-101), verbatim semantics.
// ---------------------------------------------------------------------------------------------
// imageFloatToInt, src/Trainer.cu:19-29  (x256, clamp, alpha 0xFF)
void orc_image_float_to_int(const float* src, uint32_t* fb, int w, int h) {
    const size_t N = (size_t)w * h;
    for (size_t i = 0; i < N; i++) {
        auto q = [](float v) { return (uint32_t)std::min(255, std::max(0, (int)(v * 256.0f))); };
        fb[i] = (q(src[i]) << 0) + (q(src[i + N]) << 8) + (q(src[i + 2 * N]) << 16) + (0xFFu << 24);
    }
}
// imageIntToLoss, src/Trainer.cu:33-44  (truth/255 - render)
void orc_image_int_to_loss(const uint32_t* truth, const float* rast, float* loss, int w, int h) {
    const size_t N = (size_t)w * h;
    for (size_t i = 0; i < N; i++) {
        const uint32_t t = truth[i];
        loss[i] = ((float)(t & 0xFF) / 255.0f) - rast[i];
        loss[i + N] = ((float)((t >> 8) & 0xFF) / 255.0f) - rast[i + N];
        loss[i + 2 * N] = ((float)((t >> 16) & 0xFF) / 255.0f) - rast[i + 2 * N];
    }
}
// accumulateGradients, src/Trainer.cu:47-77
void orc_accumulate(float* var, float* aLoc, float* aSh, float* aScale, float* aOpac, float* aRot, const float* gLoc,
                    const float* gSh, const float* gScale, const float* gOpac, const float* gRot, float samples, int M,
                    int n) {
    for (int i = 0; i < n; i++) {
        var[i] += sqrtf((gLoc[i * 3] * gLoc[i * 3]) + (gLoc[i * 3 + 1] * gLoc[i * 3 + 1]) + (gLoc[i * 3 + 2] * gLoc[i * 3 + 2])) / samples;
        for (int f = 0; f < 3; f++) aLoc[i * 3 + f] += gLoc[i * 3 + f] / samples;
        for (int f = 0; f < 3 * M; f++) aSh[(size_t)i * 3 * M + f] += gSh[(size_t)i * 3 * M + f] / samples;
        for (int f = 0; f < 3; f++) aScale[i * 3 + f] += gScale[i * 3 + f] / samples;
        aOpac[i] += gOpac[i] / samples;
        for (int f = 0; f < 4; f++) aRot[i * 4 + f] += gRot[i * 4 + f] / samples;
    }
}
// applyGradients, src/Trainer.cu:81-101
void orc_apply_sgd(float* loc, float* sh, float* scale, float* opac, float* rot, const float* gLoc, const float* gSh,
                   const float* gScale, const float* gOpac, const float* gRot, float lr, float lrSh, float lrScale,
                   float lrOpacity, float lrRotation, float maxScale, int M, int count) {
    for (int i = 0; i < count; i++) {
        for (int f = 0; f < 3; f++) loc[i * 3 + f] += gLoc[i * 3 + f] * lr;
        for (int f = 0; f < M * 3; f++) sh[(size_t)i * 3 * M + f] += gSh[(size_t)i * 3 * M + f] * lrSh;
        for (int f = 0; f < 3; f++) {
            scale[i * 3 + f] += gScale[i * 3 + f] * lrScale;
            scale[i * 3 + f] = std::min(maxScale, std::max(0.0f, scale[i * 3 + f]));
        }
        opac[i] = std::min(1.0f, std::max(0.0f, opac[i] + gOpac[i] * lrOpacity));
        for (int f = 0; f < 4; f++) rot[i * 4 + f] += gRot[i * 4 + f] * lrRotation;
    }
}
// Build-side extension (BASELINE.json "fwd+bwd+Adam"; the reference has no Adam):
// Adam *ascent* on the averaged residual gradient, followed by the reference's clamps.
// Layout of m / v: [loc 3P | sh 3MP | scale 3P | opac P | rot 4P], step counter t >= 1.
void orc_apply_adam(float* loc, float* sh, float* scale, float* opac, float* rot, const float* gLoc, const float* gSh,
                    const float* gScale, const float* gOpac, const float* gRot, float* m, float* v, int t, float lr,
                    float lrSh, float lrScale, float lrOpacity, float lrRotation, float maxScale, float b1, float b2,
                    float eps, int M, int count) {
    const float bc1 = 1.0f - std::pow(b1, (float)t), bc2 = 1.0f - std::pow(b2, (float)t);
    auto upd = [&](float& p, float g, float& mm, float& vv, float lrate) {
        mm = b1 * mm + (1.0f - b1) * g;
        vv = b2 * vv + (1.0f - b2) * g * g;
        const float mh = mm / bc1, vh = vv / bc2;
        p = p + lrate * (mh / (std::sqrt(vh) + eps));
    };
    const size_t P = (size_t)count;
    float *mLoc = m, *mSh = m + 3 * P, *mScale = mSh + 3 * M * P, *mOpac = mScale + 3 * P, *mRot = mOpac + P;
    float *vLoc = v, *vSh = v + 3 * P, *vScale = vSh + 3 * M * P, *vOpac = vScale + 3 * P, *vRot = vOpac + P;
    for (size_t i = 0; i < P; i++) {
        for (int f = 0; f < 3; f++) upd(loc[i * 3 + f], gLoc[i * 3 + f], mLoc[i * 3 + f], vLoc[i * 3 + f], lr);
        for (int f = 0; f < 3 * M; f++) upd(sh[i * 3 * M + f], gSh[i * 3 * M + f], mSh[i * 3 * M + f], vSh[i * 3 * M + f], lrSh);
        for (int f = 0; f < 3; f++) {
            upd(scale[i * 3 + f], gScale[i * 3 + f], mScale[i * 3 + f], vScale[i * 3 + f], lrScale);
            scale[i * 3 + f] = std::min(maxScale, std::max(0.0f, scale[i * 3 + f]));
        }
        upd(opac[i], gOpac[i], mOpac[i], vOpac[i], lrOpacity);
        opac[i] = std::min(1.0f, std::max(0.0f, opac[i]));
        for (int f = 0; f < 4; f++) upd(rot[i * 4 + f], gRot[i * 4 + f], mRot[i * 4 + f], vRot[i * 4 + f], lrRotation);
    }
}

// ---------------------------------------------------------------------------------------------
// Trainer::train, per-view loop (src/Trainer.cu:303-425): forward, loss, backward, accumulate.
// views: V records of 40 floats {view[16], projview[16], campos[3], tanfovx, tanfovy, bg[3]}
// (the values the reference builds at src/Trainer.cu:317-326,355-356).  avg*/var are accumulated
// into (the caller zeroes them as src/Trainer.cu:304-309 does).  `samples` is S = 2*#cameras.
// ---------------------------------------------------------------------------------------------
void orc_train_views(int P, int D, int M, int W, int H, int V, const float* loc, const float* sh, const float* scale,
                     const float* opac, const float* rot, const float* views, const uint32_t* truths, float samples,
                     float* var, float* aLoc, float* aSh, float* aScale, float* aOpac, float* aRot, int* num_rendered,
                     float* out_images) {
    const size_t N = (size_t)W * H;
    std::vector<float> rast(3 * N), loss(3 * N);
    std::vector<float> gLoc(3 * (size_t)P), gSh(3 * (size_t)M * P), gScale(3 * (size_t)P), gOpac(P), gRot(4 * (size_t)P);
    std::vector<float> gMean2D(3 * (size_t)P), gConic(4 * (size_t)P), gColor(3 * (size_t)P), gCov3D(6 * (size_t)P);
    orc_state* st = orc_state_new();
    for (int v = 0; v < V; v++) {
        const float* vw = views + (size_t)v * 40;
        const float *view = vw, *projview = vw + 16, *campos = vw + 32, *bg = vw + 37;
        const float tanx = vw[35], tany = vw[36];
        const int R = orc_forward_f32(st, P, D, M, bg, W, H, loc, sh, opac, scale, 1.0f, rot, view, projview, campos, tanx, tany, rast.data());
        if (num_rendered) num_rendered[v] = R;
        if (out_images) std::memcpy(out_images + (size_t)v * 3 * N, rast.data(), 3 * N * sizeof(float));
        orc_image_int_to_loss(truths + (size_t)v * N, rast.data(), loss.data(), W, H);
        auto zero = [](std::vector<float>& a) { std::fill(a.begin(), a.end(), 0.0f); };
        zero(gLoc); zero(gSh); zero(gScale); zero(gOpac); zero(gRot); zero(gMean2D); zero(gConic); zero(gColor); zero(gCov3D);
        orc_backward_f32(st, D, M, bg, loc, sh, scale, 1.0f, rot, view, projview, campos, tanx, tany, loss.data(),
                         gMean2D.data(), gConic.data(), gOpac.data(), gColor.data(), gLoc.data(), gCov3D.data(),
                         gSh.data(), gScale.data(), gRot.data(), nullptr);
        orc_accumulate(var, aLoc, aSh, aScale, aOpac, aRot, gLoc.data(), gSh.data(), gScale.data(), gOpac.data(), gRot.data(), samples, M, P);
    }
    orc_state_free(st);
}

// ---------------------------------------------------------------------------------------------
// Densify / prune, src/Trainer.cu:437-542.  The reference iterates std::unordered_set<int>
// (implementation-defined order); this restatement iterates in ascending splat index.
// quat_xyzw != 0 models glm's default quaternion member order {x,y,z,w}: glm::quat(r0,r1,r2,r3)
// is the (w,x,y,z) constructor and memcpy(&rotPre[0]) then writes {r1,r2,r3,r0} back
// (src/Trainer.cu:463,493-494); quat_xyzw == 0 models GLM_FORCE_QUAT_DATA_WXYZ (no permutation).
// Arrays are ModelSplatsHost buffers sized to `capacity`; returns the new count.
// ---------------------------------------------------------------------------------------------
// adam_m / adam_v (optional, build-side extension: the reference has no optimizer state): Adam moments in the same
// capacity-sized five-array layout [loc 3C | sh 3MC | scale 3C | opac C | rot 4C]; they follow their splat through
// every copy (both halves of a split and a clone's twin inherit the parent's moments) and the rotation rows follow
// the quaternion's member permutation.
static int densify_impl(float* loc, float* sh, float* scale, float* opac, float* rot, int count, int capacity, int M,
                        const float* var, const float* gradLoc, float cullOpacity, float cullSize, float densifyVariance,
                        float splitSize, float splitDistance, float splitScale, float cloneDistance, int quat_xyzw,
                        float* adam_m, float* adam_v) {
    std::set<int> toSplit, toClone, toRemove;
    const size_t C = (size_t)capacity;
    auto state_copy = [&](float* a, int to, int from) {
        if (!a) return;
        float *aLoc = a, *aSh = a + 3 * C, *aScale = aSh + 3 * (size_t)M * C, *aOpac = aScale + 3 * C, *aRot = aOpac + C;
        std::memcpy(&aLoc[to * 3], &aLoc[from * 3], 12);
        for (int k = 0; k < M * 3; k++) aSh[(size_t)to * 3 * M + k] = aSh[(size_t)from * 3 * M + k];
        std::memcpy(&aScale[to * 3], &aScale[from * 3], 12);
        aOpac[to] = aOpac[from];
        std::memcpy(&aRot[to * 4], &aRot[from * 4], 16);
    };
    auto state_permute_rot = [&](float* a, int i) {
        if (!a || !quat_xyzw) return;
        float* r = a + 3 * C + 3 * (size_t)M * C + 3 * C + C + (size_t)i * 4;
        const float w = r[0]; r[0] = r[1]; r[1] = r[2]; r[2] = r[3]; r[3] = w;
    };
    auto len3 = [](float a, float b, float c) { return std::sqrt(a * a + b * b + c * c); };
    for (int i = 0; i < count; i++) {
        const float sizeMag = len3(scale[i * 3], scale[i * 3 + 1], scale[i * 3 + 2]);
        if (opac[i] <= cullOpacity || sizeMag < cullSize) toRemove.insert(i);
        else if (var[i] - len3(gradLoc[i * 3], gradLoc[i * 3 + 1], gradLoc[i * 3 + 2]) > densifyVariance) {
            if (sizeMag > splitSize) toSplit.insert(i); else toClone.insert(i);
        }
    }
    auto copy = [&](int to, int from) {  // ModelSplatsHost::copy, src/ModelSplatsHost.cpp:79-91
        std::memcpy(&loc[to * 3], &loc[from * 3], 12);
        for (int k = 0; k < M * 3; k++) sh[(size_t)to * 3 * M + k] = sh[(size_t)from * 3 * M + k];
        std::memcpy(&scale[to * 3], &scale[from * 3], 12);
        opac[to] = opac[from];
        std::memcpy(&rot[to * 4], &rot[from * 4], 16);
        state_copy(adam_m, to, from); state_copy(adam_v, to, from);
    };
    for (int i : toSplit) {
        if (count >= capacity) continue;
        const float lp[3] = { loc[i * 3], loc[i * 3 + 1], loc[i * 3 + 2] };
        const float sp[3] = { scale[i * 3], scale[i * 3 + 1], scale[i * 3 + 2] };
        const float qw = rot[i * 4], qx = rot[i * 4 + 1], qy = rot[i * 4 + 2], qz = rot[i * 4 + 3];
        float so[4] = { sp[0], sp[1], sp[2], 1.0f };
        if (sp[0] > sp[1] && sp[0] > sp[2]) { so[1] *= 0.0f; so[2] *= 0.0f; }
        else if (sp[1] > sp[2]) { so[0] *= 0.0f; so[2] *= 0.0f; }
        else { so[0] *= 0.0f; so[1] *= 0.0f; }
        float Rm[3][3]; quat_to_mat3(qw, qx, qy, qz, Rm);
        // mat4(quat) * vec4: columns weighted by components, w stays 1
        const float ox = Rm[0][0] * so[0] + Rm[1][0] * so[1] + Rm[2][0] * so[2] + 0.0f * so[3];
        const float oy = Rm[0][1] * so[0] + Rm[1][1] * so[1] + Rm[2][1] * so[2] + 0.0f * so[3];
        const float oz = Rm[0][2] * so[0] + Rm[1][2] * so[1] + Rm[2][2] * so[2] + 0.0f * so[3];
        const float ow = 0.0f * so[0] + 0.0f * so[1] + 0.0f * so[2] + 1.0f * so[3];
        const float off[3] = { ox / ow, oy / ow, oz / ow };
        float n1[3], n2[3], sn[3];
        for (int k = 0; k < 3; k++) {
            n1[k] = lp[k] + off[k] * splitDistance * 0.5f;
            n2[k] = lp[k] - off[k] * splitDistance * 0.5f;
            sn[k] = sp[k] * splitScale;
        }
        const int i2 = count; count++;
        copy(i2, i);
        std::memcpy(&loc[i * 3], n1, 12); std::memcpy(&loc[i2 * 3], n2, 12);
        std::memcpy(&scale[i * 3], sn, 12); std::memcpy(&scale[i2 * 3], sn, 12);
        const float qmem_xyzw[4] = { qx, qy, qz, qw }, qmem_wxyz[4] = { qw, qx, qy, qz };
        const float* qm = quat_xyzw ? qmem_xyzw : qmem_wxyz;
        std::memcpy(&rot[i * 4], qm, 16); std::memcpy(&rot[i2 * 4], qm, 16);
        state_permute_rot(adam_m, i); state_permute_rot(adam_m, i2); state_permute_rot(adam_v, i); state_permute_rot(adam_v, i2);
    }
    for (int i : toClone) {
        if (count >= capacity) continue;
        float l[3] = { loc[i * 3], loc[i * 3 + 1], loc[i * 3 + 2] };
        const float sc[3] = { scale[i * 3], scale[i * 3 + 1], scale[i * 3 + 2] };
        const float qw = rot[i * 4], qx = rot[i * 4 + 1], qy = rot[i * 4 + 2], qz = rot[i * 4 + 3];
        const f3 dirG = norm3(f3{ gradLoc[i * 3], gradLoc[i * 3 + 1], gradLoc[i * 3 + 2] });
        float Rm[3][3]; quat_to_mat3(qw, qx, qy, qz, Rm);
        const float ox = Rm[0][0] * sc[0] + Rm[1][0] * sc[1] + Rm[2][0] * sc[2] + 0.0f * 1.0f;
        const float oy = Rm[0][1] * sc[0] + Rm[1][1] * sc[1] + Rm[2][1] * sc[2] + 0.0f * 1.0f;
        const float oz = Rm[0][2] * sc[0] + Rm[1][2] * sc[1] + Rm[2][2] * sc[2] + 0.0f * 1.0f;
        const float ow = 0.0f * sc[0] + 0.0f * sc[1] + 0.0f * sc[2] + 1.0f * 1.0f;
        l[0] += (ox / ow) * dirG.x * cloneDistance;
        l[1] += (oy / ow) * dirG.y * cloneDistance;
        l[2] += (oz / ow) * dirG.z * cloneDistance;
        const int i2 = count; count++;
        copy(i2, i);
        std::memcpy(&loc[i2 * 3], l, 12);
    }
    if (!toRemove.empty()) {
        int keep = 0;
        for (int scan = 0; scan < count; scan++)
            if (!toRemove.count(scan)) { if (keep != scan) copy(keep, scan); keep++; }
        count -= (int)toRemove.size();
    }
    return count;
}
int orc_densify(float* loc, float* sh, float* scale, float* opac, float* rot, int count, int capacity, int M,
                const float* var, const float* gradLoc, float cullOpacity, float cullSize, float densifyVariance,
                float splitSize, float splitDistance, float splitScale, float cloneDistance, int quat_xyzw) {
    return densify_impl(loc, sh, scale, opac, rot, count, capacity, M, var, gradLoc, cullOpacity, cullSize, densifyVariance, splitSize,
                        splitDistance, splitScale, cloneDistance, quat_xyzw, nullptr, nullptr);
}
int orc_densify_adam(float* loc, float* sh, float* scale, float* opac, float* rot, int count, int capacity, int M,
                     const float* var, const float* gradLoc, float cullOpacity, float cullSize, float densifyVariance,
                     float splitSize, float splitDistance, float splitScale, float cloneDistance, int quat_xyzw,
                     float* adam_m, float* adam_v) {
    return densify_impl(loc, sh, scale, opac, rot, count, capacity, M, var, gradLoc, cullOpacity, cullSize, densifyVariance, splitSize,
                        splitDistance, splitScale, cloneDistance, quat_xyzw, adam_m, adam_v);
}

// ---------------------------------------------------------------------------------------------
// Camera.cpp restatements
// ---------------------------------------------------------------------------------------------
// getFibonacciSphere, src/Camera.cpp:9-27
void orc_fibonacci_sphere(int count, float distance, float* out_xyz) {
    const float goldenRatio = (1.0f + sqrtf(5.0f)) / 2.0f;
    const float angleStep = 2.0f * (float)3.1415926535897932384626433832795 * goldenRatio;
    for (int i = 0; i < count; i++) {
        const float t = (float)i / (float)count;
        const float angle1 = acosf(1.0f - 2.0f * t);
        const float angle2 = angleStep * (float)i;
        out_xyz[3 * i] = sinf(angle1) * cosf(angle2) * distance;
        out_xyz[3 * i + 1] = sinf(angle1) * sinf(angle2) * distance;
        out_xyz[3 * i + 2] = cosf(angle1) * distance;
    }
}
// Camera::getView, src/Camera.cpp:79-82: -glm::lookAt(location, target, up=(0,1,0)) (lookAtRH)
void orc_camera_view(const float* loc, const float* target, float* out16) {
    const f3 eye = { loc[0], loc[1], loc[2] }, center = { target[0], target[1], target[2] }, up = { 0.0f, 1.0f, 0.0f };
    const f3 f = norm3(sub3(center, eye));
    const f3 s = norm3(cross3(f, up));
    const f3 u = cross3(s, f);
    float m[16] = { 0 };
    m[15] = 1.0f;
    m[0] = s.x; m[4] = s.y; m[8] = s.z;
    m[1] = u.x; m[5] = u.y; m[9] = u.z;
    m[2] = -f.x; m[6] = -f.y; m[10] = -f.z;
    m[12] = -dot3(s, eye); m[13] = -dot3(u, eye); m[14] = dot3(f, eye);
    for (int k = 0; k < 16; k++) out16[k] = -m[k];
}
// Camera::getProjection, src/Camera.cpp:84-86: glm::perspective(radians(fovY), aspect, 0.1, 100) (RH, NO)
void orc_camera_proj(float fovDegY, float aspect, float* out16) {
    const float fovy = fovDegY * 0.01745329251994329576923690768489f;
    const float zNear = 0.1f, zFar = 100.0f;
    const float tanHalfFovy = std::tan(fovy / 2.0f);
    for (int k = 0; k < 16; k++) out16[k] = 0.0f;
    out16[0] = 1.0f / (aspect * tanHalfFovy);
    out16[5] = 1.0f / (tanHalfFovy);
    out16[10] = -(zFar + zNear) / (zFar - zNear);
    out16[11] = -1.0f;
    out16[14] = -(2.0f * zFar * zNear) / (zFar - zNear);
}
// glm mat4 * mat4 (column-major): out = a * b
void orc_mat4_mul(const float* a, const float* b, float* out16) {
    float r[16];
    for (int c = 0; c < 4; c++)
        for (int rr = 0; rr < 4; rr++)
            r[c * 4 + rr] = a[0 * 4 + rr] * b[c * 4 + 0] + a[1 * 4 + rr] * b[c * 4 + 1] + a[2 * 4 + rr] * b[c * 4 + 2] + a[3 * 4 + rr] * b[c * 4 + 3];
    std::memcpy(out16, r, sizeof(r));
}

}  // extern "C"
