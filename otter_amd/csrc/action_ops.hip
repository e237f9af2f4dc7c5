// Action post-processing of a rollout step for many episodes at once, fp64 (the reference does this per episode in
// numpy float64 on the host: otter/policy/otter_interface.py:113-133, otter/dataset/utils.py:9-34,100-119,145-151,236-264,
// script/droid_inference.py:174-196).  A few hundred flops per episode: one thread per episode, no shared state.
#include "common.hpp"
#include "otter_hip.h"

namespace {

struct ActionArgs {
    const float* delta;     // [E][H][10] predicted chunks (relative to the current pose)
    const double* pose;     // [E][10] current pose: xyz, first two rows of the rotation matrix, gripper
    double* abs_chunks;     // [E][H][10] or null
    double* ring;           // [E][X][X][10] the first X actions of the last X chunks (slot = push count % X)
    int* counts;            // [E] chunks pushed since the episode's reset
    double* action10;       // [E][10] or null
    double* robot7;         // [E][7]
    int E, H, X, receding;
    double k;
};

struct Mat3 { double m[3][3]; };

// rot_6d_to_rot_mat + gram_schmidt (dataset/utils.py:19-34,100-119): rows a, b orthonormalised, third row a x b.
// (The reference skips the orthonormalisation when every row pair of its batch is orthogonal to 1e-8 and leaves the
// rest to scipy's nearest-rotation step; for such input both give the normalised rows, as this does.)
__device__ inline Mat3 rot6d_to_mat(const double* r) {
    Mat3 o;
    double a[3] = {r[0], r[1], r[2]}, b[3] = {r[3], r[4], r[5]};
    const double na = sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    for (int i = 0; i < 3; ++i) a[i] /= na;
    const double d = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
    for (int i = 0; i < 3; ++i) b[i] -= d * a[i];
    const double nb = sqrt(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
    for (int i = 0; i < 3; ++i) b[i] /= nb;
    for (int i = 0; i < 3; ++i) { o.m[0][i] = a[i]; o.m[1][i] = b[i]; }
    o.m[2][0] = a[1] * b[2] - a[2] * b[1];
    o.m[2][1] = a[2] * b[0] - a[0] * b[2];
    o.m[2][2] = a[0] * b[1] - a[1] * b[0];
    return o;
}

__global__ __launch_bounds__(64) void action_step(ActionArgs p) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= p.E) return;
    const int A = 10;
    const double* pose = p.pose + (long)e * A;
    const Mat3 rc = rot6d_to_mat(pose + 3);
    const int count = p.counts[e];
    const int slot = count % p.X;
    double* ring = p.ring + (long)e * p.X * p.X * A;
    // receding-horizon control keeps ONE chunk and refills it only when all of its X actions have been executed
    const bool store = !p.receding || slot == 0;
    // ---- convert_abs_action (dataset/utils.py:236-264): t = R_cur . dt + t_cur, R = R_cur . R_delta, gripper passes through
    for (int t = 0; t < p.H; ++t) {
        const float* d = p.delta + ((long)e * p.H + t) * A;
        double d6[6];
        for (int i = 0; i < 6; ++i) d6[i] = (double)d[3 + i];
        const Mat3 rd = rot6d_to_mat(d6);
        double out[10];
        for (int i = 0; i < 3; ++i)
            out[i] = rc.m[i][0] * (double)d[0] + rc.m[i][1] * (double)d[1] + rc.m[i][2] * (double)d[2] + pose[i];
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 3; ++j) out[3 + 3 * i + j] = rc.m[i][0] * rd.m[0][j] + rc.m[i][1] * rd.m[1][j] + rc.m[i][2] * rd.m[2][j];
        out[9] = (double)d[9];
        if (p.abs_chunks)
            for (int i = 0; i < A; ++i) p.abs_chunks[((long)e * p.H + t) * A + i] = out[i];
        if (store && t < p.X)
            for (int i = 0; i < A; ++i) ring[((long)(p.receding ? 0 : slot) * p.X + t) * A + i] = out[i];
    }
    // ---- which action is executed now (droid_inference.py:174-196)
    double act[10];
    if (p.receding) {
        for (int i = 0; i < A; ++i) act[i] = ring[(long)slot * A + i];          // element `slot` of the kept chunk
    } else {
        // temporal ensemble: chunk pushed j steps ago (j = 0 newest) predicted this step as its element j; the oldest
        // live chunk has weight exp(k*0), the newest exp(k*(live-1))
        const int live = count + 1 < p.X ? count + 1 : p.X;
        double wsum = 0.0;
        for (int i = 0; i < A; ++i) act[i] = 0.0;
        for (int idx = 0; idx < live; ++idx) {
            const int j = live - 1 - idx;                      // age of the chunk with ensemble index idx
            const int s = ((count - j) % p.X + p.X) % p.X;
            const double w = exp(p.k * (double)idx);
            wsum += w;
            for (int i = 0; i < A; ++i) act[i] += w * ring[((long)s * p.X + j) * A + i];
        }
        for (int i = 0; i < A; ++i) act[i] /= wsum;
    }
    p.counts[e] = count + 1;
    if (p.action10)
        for (int i = 0; i < A; ++i) p.action10[(long)e * A + i] = act[i];
    // ---- action_10d_to_7d (dataset/utils.py:145-151) and the gripper threshold (droid_inference.py:198-199):
    //      rotation matrix rows -> intrinsic XYZ Euler angles, R = Rx(a) Ry(b) Rz(c)
    const Mat3 r = rot6d_to_mat(act + 3);
    double* o = p.robot7 + (long)e * 7;
    o[0] = act[0]; o[1] = act[1]; o[2] = act[2];
    const double sb = fmin(1.0, fmax(-1.0, r.m[0][2]));
    o[3] = atan2(-r.m[1][2], r.m[2][2]);
    o[4] = asin(sb);
    o[5] = atan2(-r.m[0][1], r.m[0][0]);
    o[6] = act[9] > 0.5 ? 1.0 : 0.0;
}

}  // namespace

extern "C" int otter_action_step(const float* delta_chunks, const double* cur_pose, double* abs_chunks, double* ring, int* counts,
                                 double* action10, double* robot7, int n_episodes, int horizon, int action_dim, int exec_horizon,
                                 int receding, double k, void* stream) {
    OTTER_REQUIRE(delta_chunks && cur_pose && ring && counts && robot7, OTTER_ERR_INVALID_ARGUMENT, "otter_action_step: null operand");
    if (n_episodes == 0) return OTTER_OK;
    OTTER_REQUIRE(n_episodes > 0 && horizon > 0, OTTER_ERR_INVALID_ARGUMENT, "otter_action_step: n_episodes=%d horizon=%d", n_episodes, horizon);
    OTTER_REQUIRE(action_dim == 10, OTTER_ERR_UNSUPPORTED_SHAPE,
                  "otter_action_step: action_dim=%d; the pose layout xyz + rot6d + gripper has 10 entries", action_dim);
    OTTER_REQUIRE(exec_horizon >= 1 && exec_horizon <= horizon, OTTER_ERR_INVALID_ARGUMENT,
                  "otter_action_step: exec_horizon=%d must be in 1..horizon=%d", exec_horizon, horizon);
    ActionArgs a{delta_chunks, cur_pose, abs_chunks, ring, counts, action10, robot7, n_episodes, horizon, exec_horizon, receding ? 1 : 0, k};
    hipLaunchKernelGGL(action_step, dim3((n_episodes + 63) / 64), dim3(64), 0, (hipStream_t)stream, a);
    OTTER_CHECK_LAUNCH("otter_action_step");
    return OTTER_OK;
}
