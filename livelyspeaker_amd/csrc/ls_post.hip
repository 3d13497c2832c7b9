// Caller-side post-processing of a sampled batch, one workgroup per clip:
//   aligned_motions = sample.permute(0,3,1,2).reshape(B,34,27)                      scripts/test_RAG_ted.py:84
//   beat_vec = normalize(aligned + mean_dir_vec);  joint-angle change curve;  motion beats   :88-111
//   convert_dir_vec_to_pose(aligned + mean_dir_vec)                                scripts/utils/data_utils.py:77-97
// Tiny and latency-bound (918 floats per clip); it exists so the sampled tensor never has to leave the GPU in the
// reference's [B,J,F,T] layout just to be transposed and scanned on the host.
#include "ls_hip.h"
#include "ls_internal.h"

namespace ls {

constexpr int kMaxBones = 16, kMaxPairs = 8;

struct PostParams {
    int njoints;                       // bones (direction vectors), 9 for TED
    int n_pairs;
    int pair_a[kMaxPairs], pair_b[kMaxPairs];
    float change_angle[kMaxPairs];
    float thres;
    int n_pose_joints;                 // 10
    int bone_parent[kMaxBones], bone_child[kMaxBones];
    float bone_len[kMaxBones];
    float mean_dir_vec[kMaxBones * 3];
};

__global__ __launch_bounds__(64) void k_ted_post(const float* __restrict__ sample, PostParams p, float* __restrict__ aligned,
                                                 float* __restrict__ pose, float* __restrict__ angle_diff,
                                                 unsigned char* __restrict__ beat_mask) {
    __shared__ float sv[kT][kMaxBones * 3];     // aligned + mean (un-normalised)
    __shared__ float sn[kT][kMaxBones * 3];     // per-bone unit vectors
    __shared__ float sang[kMaxPairs][kT];
    __shared__ float sdiff[kT];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int JF = p.njoints * 3;
    for (int i = tid; i < kT * JF; i += 64) {
        const int f = i / JF, c = i - f * JF;
        const float v = sample[((size_t)b * JF + c) * kT + f];          // [B,J,F,T] -> [B,T,J*F]
        if (aligned) aligned[(size_t)b * kT * JF + i] = v;
        sv[f][c] = v + p.mean_dir_vec[c];
    }
    __syncthreads();
    for (int i = tid; i < kT * p.njoints; i += 64) {                    // F.normalize(dim=-1): x / max(||x||, 1e-12)
        const int f = i / p.njoints, j = i - f * p.njoints;
        const float x = sv[f][3 * j], y = sv[f][3 * j + 1], z = sv[f][3 * j + 2];
        const float inv = 1.0f / fmaxf(sqrtf(x * x + y * y + z * z), 1e-12f);
        sn[f][3 * j] = x * inv; sn[f][3 * j + 1] = y * inv; sn[f][3 * j + 2] = z * inv;
    }
    __syncthreads();
    for (int i = tid; i < kT * p.n_pairs; i += 64) {                    // angle between the two bones of each pair
        const int k = i / kT, f = i - k * kT;
        const float* u = &sn[f][3 * p.pair_a[k]];
        const float* v = &sn[f][3 * p.pair_b[k]];
        float ip = u[0] * v[0] + u[1] * v[1] + u[2] * v[2];
        ip = fminf(fmaxf(ip, -1.0f), 1.0f);
        sang[k][f] = acosf(ip) * 0.3183098861837907f;
    }
    __syncthreads();
    if (tid < kT) {
        float d = 0.f;
        if (tid > 0)
            for (int k = 0; k < p.n_pairs; ++k)
                d += fabsf(sang[k][tid] - sang[k][tid - 1]) / p.change_angle[k] / (float)p.n_pairs;
        sdiff[tid] = d;
        if (angle_diff) angle_diff[(size_t)b * kT + tid] = d;
    }
    __syncthreads();
    if (beat_mask && tid < kT) {                                        // local minima of the change curve, t in [2, 32]
        bool beat = false;
        if (tid >= 2 && tid <= kT - 2) {
            const float c = sdiff[tid], l = sdiff[tid - 1], r = sdiff[tid + 1];
            beat = (c < l && c < r) && (l - c >= p.thres || r - c >= p.thres);
        }
        beat_mask[(size_t)b * kT + tid] = beat ? 1 : 0;
    }
    if (pose) {                                                         // joint positions along the bone tree
        for (int f = tid; f < kT; f += 64) {
            float jp[kMaxBones + 1][3];
            for (int j = 0; j < p.n_pose_joints; ++j) jp[j][0] = jp[j][1] = jp[j][2] = 0.f;
            for (int j = 0; j < p.njoints; ++j)
                for (int e = 0; e < 3; ++e) jp[p.bone_child[j]][e] = jp[p.bone_parent[j]][e] + p.bone_len[j] * sv[f][3 * j + e];
            float* o = pose + ((size_t)b * kT + f) * p.n_pose_joints * 3;
            for (int j = 0; j < p.n_pose_joints; ++j)
                for (int e = 0; e < 3; ++e) o[3 * j + e] = jp[j][e];
        }
    }
}

}  // namespace ls

extern "C" int ls_ted_post(int device, int on_device, int batch, const ls_post_config* c, const float* sample,
                           float* aligned, float* pose, float* angle_diff, unsigned char* beat_mask) {
    using namespace ls;
    if (!c || !sample || batch < 1) return LS_EINVAL;
    if (c->njoints < 1 || c->njoints > kMaxBones || c->n_pairs < 0 || c->n_pairs > kMaxPairs || c->n_pose_joints > kMaxBones + 1)
        return LS_EINVAL;
    if (hipSetDevice(device) != hipSuccess) return LS_EHIP;
    PostParams p{};
    p.njoints = c->njoints; p.n_pairs = c->n_pairs; p.thres = c->thres; p.n_pose_joints = c->n_pose_joints;
    for (int k = 0; k < c->n_pairs; ++k) { p.pair_a[k] = c->pair_a[k]; p.pair_b[k] = c->pair_b[k]; p.change_angle[k] = c->change_angle[k]; }
    for (int j = 0; j < c->njoints; ++j) { p.bone_parent[j] = c->bone_parent[j]; p.bone_child[j] = c->bone_child[j]; p.bone_len[j] = c->bone_len[j]; }
    for (int j = 0; j < c->njoints * 3; ++j) p.mean_dir_vec[j] = c->mean_dir_vec[j];
    const int JF = c->njoints * 3;
    const size_t n_in = (size_t)batch * JF * kT, n_pose = (size_t)batch * kT * c->n_pose_joints * 3, n_t = (size_t)batch * kT;
    float *d_in = nullptr, *d_al = nullptr, *d_pose = nullptr, *d_diff = nullptr;
    unsigned char* d_mask = nullptr;
    hipError_t e = hipSuccess;
    auto chk = [&](hipError_t x) { if (e == hipSuccess) e = x; };
    if (on_device) {
        d_in = const_cast<float*>(sample); d_al = aligned; d_pose = pose; d_diff = angle_diff; d_mask = beat_mask;
    } else {
        chk(hipMalloc(&d_in, n_in * 4));
        if (aligned) chk(hipMalloc(&d_al, n_in * 4));
        if (pose) chk(hipMalloc(&d_pose, n_pose * 4));
        if (angle_diff) chk(hipMalloc(&d_diff, n_t * 4));
        if (beat_mask) chk(hipMalloc(&d_mask, n_t));
        if (e == hipSuccess) chk(hipMemcpy(d_in, sample, n_in * 4, hipMemcpyHostToDevice));
    }
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_ted_post, dim3(batch), dim3(64), 0, 0, d_in, p, d_al, d_pose, d_diff, d_mask);
        chk(hipGetLastError());
        chk(hipDeviceSynchronize());
    }
    if (!on_device) {
        if (e == hipSuccess && aligned) chk(hipMemcpy(aligned, d_al, n_in * 4, hipMemcpyDeviceToHost));
        if (e == hipSuccess && pose) chk(hipMemcpy(pose, d_pose, n_pose * 4, hipMemcpyDeviceToHost));
        if (e == hipSuccess && angle_diff) chk(hipMemcpy(angle_diff, d_diff, n_t * 4, hipMemcpyDeviceToHost));
        if (e == hipSuccess && beat_mask) chk(hipMemcpy(beat_mask, d_mask, n_t, hipMemcpyDeviceToHost));
        (void)hipFree(d_in); if (d_al) (void)hipFree(d_al); if (d_pose) (void)hipFree(d_pose);
        if (d_diff) (void)hipFree(d_diff); if (d_mask) (void)hipFree(d_mask);
    }
    return e == hipSuccess ? LS_OK : LS_EHIP;
}

// ---- BEAT: rot6d -> rotation matrix -> Euler XYZ (degrees), plus the [B,J,6,T] -> [B,T,J*6] layout change --------------------
namespace ls {

__global__ __launch_bounds__(256) void k_beat_post(const float* __restrict__ sample, float* __restrict__ decoded, float* __restrict__ euler,
                                                   int J, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;          // one thread per (b, t, joint)
    if (i >= total) return;
    const int j = (int)(i % J);
    const size_t bt = i / J;
    const int t = (int)(bt % kT);
    const size_t b = bt / kT;
    float d6[6];
#pragma unroll
    for (int f = 0; f < 6; ++f) d6[f] = sample[((b * J + j) * 6 + f) * kT + t];
    if (decoded) {
#pragma unroll
        for (int f = 0; f < 6; ++f) decoded[(bt * J + j) * 6 + f] = d6[f];
    }
    if (!euler) return;
    // rotation_6d_to_matrix (rot_utils.py:529-534): b1 = normalize(a1), b2 = normalize(a2 - (b1.a2) b1), b3 = b1 x b2; rows of M
    const float n1 = fmaxf(sqrtf(d6[0] * d6[0] + d6[1] * d6[1] + d6[2] * d6[2]), 1e-12f);      // F.normalize: x / max(|x|, eps)
    const float b1x = d6[0] / n1, b1y = d6[1] / n1, b1z = d6[2] / n1;
    const float dt = b1x * d6[3] + b1y * d6[4] + b1z * d6[5];
    float b2x = d6[3] - dt * b1x, b2y = d6[4] - dt * b1y, b2z = d6[5] - dt * b1z;
    const float n2 = fmaxf(sqrtf(b2x * b2x + b2y * b2y + b2z * b2z), 1e-12f);
    b2x /= n2; b2y /= n2; b2z /= n2;
    const float b3z = b1x * b2y - b1y * b2x;                           // only m22 of the third row is needed
    // matrix_to_euler_angles(M, "XYZ") (rot_utils.py:238-257): (atan2(-m12, m22), asin(m02), atan2(-m01, m00))
    const float k = 57.29577951308232f;                                // / pi * 180
    float* o = euler + (bt * J + j) * 3;
    o[0] = atan2f(-b2z, b3z) * k;
    o[1] = asinf(b1z) * k;
    o[2] = atan2f(-b1y, b1x) * k;
}

}  // namespace ls

extern "C" int ls_beat_post(int device, int on_device, int batch, int njoints, const float* sample, float* decoded, float* euler_deg) {
    using namespace ls;
    if (!sample || batch < 1 || njoints < 1) return LS_EINVAL;
    if (hipSetDevice(device) != hipSuccess) return LS_EHIP;
    const size_t n_in = (size_t)batch * njoints * 6 * kT, n_eu = (size_t)batch * kT * njoints * 3, total = (size_t)batch * kT * njoints;
    float *d_in = nullptr, *d_dec = nullptr, *d_eu = nullptr;
    hipError_t e = hipSuccess;
    auto chk = [&](hipError_t x) { if (e == hipSuccess) e = x; };
    if (on_device) {
        d_in = const_cast<float*>(sample); d_dec = decoded; d_eu = euler_deg;
    } else {
        chk(hipMalloc(&d_in, n_in * 4));
        if (decoded) chk(hipMalloc(&d_dec, n_in * 4));
        if (euler_deg) chk(hipMalloc(&d_eu, n_eu * 4));
        if (e == hipSuccess) chk(hipMemcpy(d_in, sample, n_in * 4, hipMemcpyHostToDevice));
    }
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_beat_post, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, d_in, d_dec, d_eu, njoints, total);
        chk(hipGetLastError());
        chk(hipDeviceSynchronize());
    }
    if (!on_device) {
        if (e == hipSuccess && decoded) chk(hipMemcpy(decoded, d_dec, n_in * 4, hipMemcpyDeviceToHost));
        if (e == hipSuccess && euler_deg) chk(hipMemcpy(euler_deg, d_eu, n_eu * 4, hipMemcpyDeviceToHost));
        (void)hipFree(d_in); if (d_dec) (void)hipFree(d_dec); if (d_eu) (void)hipFree(d_eu);
    }
    return e == hipSuccess ? LS_OK : LS_EHIP;
}
