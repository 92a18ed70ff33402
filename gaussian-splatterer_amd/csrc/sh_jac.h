// sh_jac.h — d(colour channel)/d(view direction) of the SH expansion, one source for the two places that need it.
// The per-splat backward needs, for every channel, (dx, dy, dz) = sum_k sh[k][ch] * d basis_k / d(X, Y, Z)
// (upstream computeColorFromSH backward).  It depends on the camera and the splat but not on
// the pass, and the projection kernel already has all 3*M SH coefficients in flight for the colour: the trainer
// computes it there (k_preprocess.hip) and the per-(pass, splat) backward reads 9 floats instead of 3*M coefficients
// (16 passes x 100k splats x 192 B at cfg3).  Same expression text, same -ffp-contract=off build => same bits as
// evaluating it in the backward (the rasterizer seam still does that).
#pragma once

namespace gs {

constexpr float SHJ_C1 = 0.4886025119029199f;
constexpr float SHJ_C2_0 = 1.0925484305920792f, SHJ_C2_1 = -1.0925484305920792f, SHJ_C2_2 = 0.31539156525252005f,
                SHJ_C2_3 = -1.0925484305920792f, SHJ_C2_4 = 0.5462742152960396f;
constexpr float SHJ_C3_0 = -0.5900435899266435f, SHJ_C3_1 = 2.890611442640554f, SHJ_C3_2 = -0.4570457994644658f,
                SHJ_C3_3 = 0.3731763325901154f, SHJ_C3_4 = -0.4570457994644658f, SHJ_C3_5 = 1.445305721320277f,
                SHJ_C3_6 = -0.5900435899266435f;

// sh(k) returns coefficient k of the channel in question
template <int D, class ShOfChannel>
__device__ inline void sh_direction_jacobian(float X, float Y, float Z, ShOfChannel sh, float& dx_, float& dy_, float& dz_) {
    dx_ = 0; dy_ = 0; dz_ = 0;
    if constexpr (D > 0) { dx_ = -SHJ_C1 * sh(3); dy_ = -SHJ_C1 * sh(1); dz_ = SHJ_C1 * sh(2); }
    if constexpr (D > 1) {
        const float xx = X * X, yy = Y * Y, zz = Z * Z, xy = X * Y, yz = Y * Z, xz = X * Z;
        dx_ += SHJ_C2_0 * Y * sh(4) + SHJ_C2_2 * 2.0f * -X * sh(6) + SHJ_C2_3 * Z * sh(7) + SHJ_C2_4 * 2.0f * X * sh(8);
        dy_ += SHJ_C2_0 * X * sh(4) + SHJ_C2_1 * Z * sh(5) + SHJ_C2_2 * 2.0f * -Y * sh(6) + SHJ_C2_4 * 2.0f * -Y * sh(8);
        dz_ += SHJ_C2_1 * Y * sh(5) + SHJ_C2_2 * 2.0f * 2.0f * Z * sh(6) + SHJ_C2_3 * X * sh(7);
        if constexpr (D > 2) {
            dx_ += SHJ_C3_0 * sh(9) * 3.0f * 2.0f * xy + SHJ_C3_1 * sh(10) * yz + SHJ_C3_2 * sh(11) * -2.0f * xy +
                   SHJ_C3_3 * sh(12) * -3.0f * 2.0f * xz + SHJ_C3_4 * sh(13) * (-3.0f * xx + 4.0f * zz - yy) +
                   SHJ_C3_5 * sh(14) * 2.0f * xz + SHJ_C3_6 * sh(15) * 3.0f * (xx - yy);
            dy_ += SHJ_C3_0 * sh(9) * 3.0f * (xx - yy) + SHJ_C3_1 * sh(10) * xz +
                   SHJ_C3_2 * sh(11) * (-3.0f * yy + 4.0f * zz - xx) + SHJ_C3_3 * sh(12) * -3.0f * 2.0f * yz +
                   SHJ_C3_4 * sh(13) * -2.0f * xy + SHJ_C3_5 * sh(14) * -2.0f * yz + SHJ_C3_6 * sh(15) * -3.0f * 2.0f * xy;
            dz_ += SHJ_C3_1 * sh(10) * xy + SHJ_C3_2 * sh(11) * 4.0f * 2.0f * yz +
                   SHJ_C3_3 * sh(12) * 3.0f * (2.0f * zz - xx - yy) + SHJ_C3_4 * sh(13) * 4.0f * 2.0f * xz +
                   SHJ_C3_5 * sh(14) * (xx - yy);
        }
    }
}

}  // namespace gs
