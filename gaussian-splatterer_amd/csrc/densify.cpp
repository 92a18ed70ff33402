// densify.cpp — split / clone / prune on the host, as the reference does (src/Trainer.cu:433-542:
// "This is done on the CPU instead of GPU ... it's only about 1/200 iterations").
//
// Semantics kept from the reference:
//   cull   : opacity <= cullOpacity  or  |scale| < cullSize                       (:451)
//   densify: var - |avgGradLoc| > densifyVariance; split if |scale| > splitSize else clone (:453-454)
//   split  : copies move +-0.5*splitDistance along the largest scale axis rotated by q, scale *= splitScale (:459-496)
//   clone  : copy offset by (R*scale) (.) normalize(avgGradLoc) * cloneDistance    (:499-521)
//   prune  : stable compaction in index order, count -= |toRemove|               (:524-534)
// Deviation, documented: the reference walks std::unordered_set<int> (implementation-defined
// order); here candidates are walked in ascending splat index, which makes the result portable.
#include <cmath>
#include <cstring>
#include <vector>

#include "gs_internal.h"

namespace gs {

namespace {
struct Mat3 { float m[3][3]; };  // m[col][row], glm::mat3_cast(quat(w,x,y,z)) without normalisation
Mat3 rotation_of(float w, float x, float y, float z) {
    Mat3 R;
    R.m[0][0] = 1.0f - 2.0f * (y * y + z * z); R.m[0][1] = 2.0f * (x * y + w * z); R.m[0][2] = 2.0f * (x * z - w * y);
    R.m[1][0] = 2.0f * (x * y - w * z); R.m[1][1] = 1.0f - 2.0f * (x * x + z * z); R.m[1][2] = 2.0f * (y * z + w * x);
    R.m[2][0] = 2.0f * (x * z + w * y); R.m[2][1] = 2.0f * (y * z - w * x); R.m[2][2] = 1.0f - 2.0f * (x * x + y * y);
    return R;
}
// (mat4)q * vec4(v, 1) followed by the reference's divide by w (w is exactly 1)
void rotate(const Mat3& R, const float v[3], float out[3]) {
    const float w = 0.0f * v[0] + 0.0f * v[1] + 0.0f * v[2] + 1.0f * 1.0f;
    for (int r = 0; r < 3; r++) out[r] = (R.m[0][r] * v[0] + R.m[1][r] * v[1] + R.m[2][r] * v[2] + 0.0f * 1.0f) / w;
}
float norm3(const float* v) { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
}  // namespace

int densify_host(float* loc, float* sh, float* scale, float* opac, float* rot, int count, int capacity, int M,
                 const float* var, const float* grad_loc, const gs_hyper& h) {
    enum : unsigned char { KEEP = 0, SPLIT = 1, CLONE = 2, REMOVE = 3 };
    std::vector<unsigned char> action((size_t)count, KEEP);
    int n_remove = 0;
    for (int i = 0; i < count; i++) {
        const float size = norm3(&scale[3 * (size_t)i]);
        if (opac[i] <= h.cull_opacity || size < h.cull_size) { action[i] = REMOVE; n_remove++; }
        else if (var[i] - norm3(&grad_loc[3 * (size_t)i]) > h.densify_variance) action[i] = size > h.split_size ? SPLIT : CLONE;
    }
    const int original = count;
    auto duplicate = [&](int from) {  // ModelSplatsHost::copy(count, from) after count++ (src/ModelSplatsHost.cpp:79-91)
        const int to = count++;
        std::memcpy(&loc[3 * (size_t)to], &loc[3 * (size_t)from], 12);
        std::memcpy(&sh[(size_t)to * 3 * M], &sh[(size_t)from * 3 * M], sizeof(float) * 3 * M);
        std::memcpy(&scale[3 * (size_t)to], &scale[3 * (size_t)from], 12);
        opac[to] = opac[from];
        std::memcpy(&rot[4 * (size_t)to], &rot[4 * (size_t)from], 16);
        return to;
    };
    for (int i = 0; i < original; i++) {
        if (action[i] != SPLIT || count >= capacity) continue;
        float* s = &scale[3 * (size_t)i];
        float* q = &rot[4 * (size_t)i];
        float axis[3] = { s[0], s[1], s[2] };
        if (s[0] > s[1] && s[0] > s[2]) { axis[1] *= 0.0f; axis[2] *= 0.0f; }
        else if (s[1] > s[2]) { axis[0] *= 0.0f; axis[2] *= 0.0f; }
        else { axis[0] *= 0.0f; axis[1] *= 0.0f; }
        float off[3];
        rotate(rotation_of(q[0], q[1], q[2], q[3]), axis, off);
        const float centre[3] = { loc[3 * (size_t)i], loc[3 * (size_t)i + 1], loc[3 * (size_t)i + 2] };
        const float shrunk[3] = { s[0] * h.split_scale, s[1] * h.split_scale, s[2] * h.split_scale };
        // glm::quat(r0,r1,r2,r3) is the (w,x,y,z) constructor; memcpy(&q[0]) then stores glm's member order
        const float stored_xyzw[4] = { q[1], q[2], q[3], q[0] };
        const float stored_wxyz[4] = { q[0], q[1], q[2], q[3] };
        const float* stored = h.quat_layout == GS_QUAT_XYZW ? stored_xyzw : stored_wxyz;
        const int twin = duplicate(i);
        for (int k = 0; k < 3; k++) {
            loc[3 * (size_t)i + k] = centre[k] + off[k] * h.split_distance * 0.5f;
            loc[3 * (size_t)twin + k] = centre[k] - off[k] * h.split_distance * 0.5f;
            scale[3 * (size_t)i + k] = shrunk[k];
            scale[3 * (size_t)twin + k] = shrunk[k];
        }
        std::memcpy(&rot[4 * (size_t)i], stored, 16);
        std::memcpy(&rot[4 * (size_t)twin], stored, 16);
    }
    for (int i = 0; i < original; i++) {
        if (action[i] != CLONE || count >= capacity) continue;
        const float* q = &rot[4 * (size_t)i];
        const float* g = &grad_loc[3 * (size_t)i];
        const float inv = 1.0f / norm3(g);  // glm::normalize = v * inversesqrt(dot(v,v))
        const float dir[3] = { g[0] * inv, g[1] * inv, g[2] * inv };
        float off[3];
        rotate(rotation_of(q[0], q[1], q[2], q[3]), &scale[3 * (size_t)i], off);
        const int twin = duplicate(i);
        for (int k = 0; k < 3; k++) loc[3 * (size_t)twin + k] = loc[3 * (size_t)i + k] + off[k] * dir[k] * h.clone_distance;
    }
    if (n_remove > 0) {
        int keep = 0;
        for (int scan = 0; scan < count; scan++) {
            if (scan < original && action[scan] == REMOVE) continue;
            if (keep != scan) {
                std::memcpy(&loc[3 * (size_t)keep], &loc[3 * (size_t)scan], 12);
                std::memmove(&sh[(size_t)keep * 3 * M], &sh[(size_t)scan * 3 * M], sizeof(float) * 3 * M);
                std::memcpy(&scale[3 * (size_t)keep], &scale[3 * (size_t)scan], 12);
                opac[keep] = opac[scan];
                std::memcpy(&rot[4 * (size_t)keep], &rot[4 * (size_t)scan], 16);
            }
            keep++;
        }
        count -= n_remove;
    }
    return count;
}

}  // namespace gs
