// geom_kernels.hip -- K8: keypoint depth -> world landmarks and the device glue around it.
//
//  * find3d_disparity_kernel : Frame::find_3d (/root/reference/src/stereo_visual_slam_main/types_def.cpp:9-18) +
//                              the gates of VO::set_ref_3d_position (visual_odometry.cpp:176-217) on a disparity map.
//  * triangulate_kernel      : north_star stage K8 -- rectified-stereo inhomogeneous DLT on matched (uL,vL),(uR,vR)
//                              with the same gates/outputs (the reference gets depth from cv::StereoSGBM instead).
//  * gather_uv_kernel        : matched keypoint coordinates -> SoA (coalesced 8-B stores).
// All f64 math, f32 at rest (cv::Point3f, quirk Q4).  Elementwise, HBM-bound: 16 B in + 14 B out per match.
#include "vslam_internal.h"

#include "se3_device.h"

namespace vslam {

__device__ inline void gate_store(const double rel[3], const double Rinv[9], const double tinv[3], const CamParams& cam, size_t i,
                                  float* xyz, uint8_t* valid, uint8_t* reliable) {
    const double X = rel[0], Y = rel[1], Z = rel[2];
    const double wx = Rinv[0] * X + Rinv[1] * Y + Rinv[2] * Z + tinv[0];
    const double wy = Rinv[3] * X + Rinv[4] * Y + Rinv[5] * Z + tinv[1];
    const double wz = Rinv[6] * X + Rinv[7] * Y + Rinv[8] * Z + tinv[2];
    const bool ok = (Z > cam.dmin && Z < cam.dmax);  // visual_odometry.cpp:194
    valid[i] = ok;
    reliable[i] = ok && Z < cam.drel;                // :201
    xyz[3 * i] = (float)wx; xyz[3 * i + 1] = (float)wy; xyz[3 * i + 2] = (float)wz;
}

__device__ inline void inverse_pose(const double* T, double Rinv[9], double tinv[3]) {
    double R[9];
    se3::rotmat(T, R);
    // T^-1 = (R^T, -R^T t)
    Rinv[0] = R[0]; Rinv[1] = R[3]; Rinv[2] = R[6];
    Rinv[3] = R[1]; Rinv[4] = R[4]; Rinv[5] = R[7];
    Rinv[6] = R[2]; Rinv[7] = R[5]; Rinv[8] = R[8];
    for (int a = 0; a < 3; ++a) tinv[a] = -(Rinv[3 * a] * T[4] + Rinv[3 * a + 1] * T[5] + Rinv[3 * a + 2] * T[6]);
}

__global__ __launch_bounds__(256) void find3d_disparity_kernel(const vslam_keypoint* __restrict__ kps, int n, const float* __restrict__ disp,
                                                              int w, int h, int dstride, const double* __restrict__ T, CamParams cam,
                                                              float* __restrict__ xyz, uint8_t* __restrict__ valid, uint8_t* __restrict__ rel) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double Rinv[9], tinv[3];
    inverse_pose(T, Rinv, tinv);
    const float u = kps[i].x, v = kps[i].y;
    const int r = (int)v, c = (int)u; // at<float>(kp.pt.y, kp.pt.x): float -> int truncation (quirk Q3)
    if (r < 0 || r >= h || c < 0 || c >= w) { valid[i] = 0; rel[i] = 0; xyz[3 * i] = xyz[3 * i + 1] = xyz[3 * i + 2] = 0.f; return; }
    const double x = ((double)u - cam.cx) / cam.fx, y = ((double)v - cam.cy) / cam.fy;
    const double depth = cam.fx * cam.b / (double)disp[(size_t)r * dstride + c];
    const double p[3] = {x * depth, y * depth, depth};
    gate_store(p, Rinv, tinv, cam, i, xyz, valid, rel);
}

// batched form: item b has d_n[b] keypoints at kps + b * kp_capacity and its own disparity map and pose
__global__ __launch_bounds__(256) void find3d_disparity_batch_kernel(const vslam_keypoint* __restrict__ kps, const int32_t* __restrict__ d_n,
                                                                    int kp_capacity, const float* __restrict__ disp, int w, int h,
                                                                    const double* __restrict__ d_T, CamParams cam, float* __restrict__ xyz,
                                                                    uint8_t* __restrict__ valid, uint8_t* __restrict__ rel) {
    const int b = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
    if (k >= min(d_n[b], kp_capacity)) return;
    const size_t i = (size_t)b * kp_capacity + k;
    double Rinv[9], tinv[3];
    inverse_pose(d_T + 7 * b, Rinv, tinv);
    const float u = kps[i].x, v = kps[i].y;
    const int r = (int)v, c = (int)u; // quirk Q3: truncation
    if (r < 0 || r >= h || c < 0 || c >= w) { valid[i] = 0; rel[i] = 0; xyz[3 * i] = xyz[3 * i + 1] = xyz[3 * i + 2] = 0.f; return; }
    const double x = ((double)u - cam.cx) / cam.fx, y = ((double)v - cam.cy) / cam.fy;
    const double depth = cam.fx * cam.b / (double)disp[((size_t)b * h + r) * w + c];
    const double p[3] = {x * depth, y * depth, depth};
    gate_store(p, Rinv, tinv, cam, i, xyz, valid, rel);
}

__global__ __launch_bounds__(256) void triangulate_kernel(const float* __restrict__ uvL, const float* __restrict__ uvR,
                                                         const int32_t* __restrict__ d_n, int capacity, const double* __restrict__ d_T,
                                                         CamParams cam, float* __restrict__ xyz, uint8_t* __restrict__ valid,
                                                         uint8_t* __restrict__ rel) {
    const int b = blockIdx.y;
    const int n = min(d_n[b], capacity);
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const size_t i = (size_t)b * capacity + k;
    double Rinv[9], tinv[3];
    inverse_pose(d_T + 7 * b, Rinv, tinv);
    const float2 l = reinterpret_cast<const float2*>(uvL)[i], r = reinterpret_cast<const float2*>(uvR)[i];
    const double fx = cam.fx, fy = cam.fy, bl = cam.b;
    const double aL = (double)l.x - cam.cx, bL = (double)l.y - cam.cy, aR = (double)r.x - cam.cx, bR = (double)r.y - cam.cy;
    // rows: [fx 0 -aL | 0], [0 fy -bL | 0], [fx 0 -aR | fx b], [0 fy -bR | 0]  -> normal equations, X and Y eliminated
    const double n00 = 2 * fx * fx, n11 = 2 * fy * fy;
    const double n02 = -fx * (aL + aR), n12 = -fy * (bL + bR);
    const double n22 = aL * aL + bL * bL + aR * aR + bR * bR;
    const double r0 = fx * fx * bl, r2 = -aR * fx * bl;
    const double s22 = n22 - n02 * n02 / n00 - n12 * n12 / n11;
    const double s2 = r2 - n02 * r0 / n00;
    const double Z = s2 / s22;
    double p[3] = {(r0 - n02 * Z) / n00, (0.0 - n12 * Z) / n11, Z};
    if (!(s22 > 0) || !isfinite(Z)) { p[0] = p[1] = 0; p[2] = -1; }
    // epipolar gate of a rectified pair (a cross-checked descriptor match has no row constraint of its own; SGBM, the
    // reference's depth source, searches along the row by construction): same row within row_tol px, positive disparity
    if (cam.row_tol >= 0 && (!(fabs((double)l.y - (double)r.y) <= cam.row_tol) || !(l.x > r.x))) { p[0] = p[1] = 0; p[2] = -1; }
    gate_store(p, Rinv, tinv, cam, i, xyz, valid, rel);
}

__global__ __launch_bounds__(256) void gather_uv_kernel(const vslam_keypoint* __restrict__ kpsQ, const vslam_keypoint* __restrict__ kpsT,
                                                       int kp_capacity, const vslam_dmatch* __restrict__ m, const int32_t* __restrict__ nm,
                                                       int match_capacity, float* __restrict__ uvQ, float* __restrict__ uvT) {
    const int b = blockIdx.y;
    const int n = min(nm[b], match_capacity);
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const size_t i = (size_t)b * match_capacity + k;
    const vslam_dmatch mm = m[i];
    const int qi = min(max(mm.queryIdx, 0), kp_capacity - 1), ti = min(max(mm.trainIdx, 0), kp_capacity - 1);
    const vslam_keypoint* q = kpsQ + (size_t)b * kp_capacity + qi;
    const vslam_keypoint* t = kpsT + (size_t)b * kp_capacity + ti;
    reinterpret_cast<float2*>(uvQ)[i] = make_float2(q->x, q->y);
    reinterpret_cast<float2*>(uvT)[i] = make_float2(t->x, t->y);
}

// One workgroup per item.  Phase 1: kp2lr[queryIdx of L/R match] = L/R match index (or -1).  Phase 2: walk the
// frame-to-frame matches in order; a match whose query keypoint has a valid triangulated point contributes
// (xyz of that point, pixel of the train keypoint).  Ordered compaction via ballot ranks.
__global__ __launch_bounds__(256) void build_pnp_inputs_kernel(const vslam_dmatch* __restrict__ d_m, const int32_t* __restrict__ d_nm,
                                                              int match_capacity, const vslam_dmatch* __restrict__ d_lr,
                                                              const int32_t* __restrict__ d_nlr, int lr_capacity,
                                                              const float* __restrict__ d_xyz_lr, const uint8_t* __restrict__ d_valid_lr,
                                                              const vslam_keypoint* __restrict__ d_kpsT, int kp_capacity,
                                                              int32_t* __restrict__ d_kp2lr, float* __restrict__ d_xyz_out,
                                                              float* __restrict__ d_uv_out, int32_t* __restrict__ d_nout, int out_capacity) {
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ int s_tot[4];
    int32_t* kp2lr = d_kp2lr + (size_t)b * kp_capacity;
    for (int i = tid; i < kp_capacity; i += 256) kp2lr[i] = -1;
    __syncthreads();
    const int nlr = min(d_nlr[b], lr_capacity);
    const vslam_dmatch* lr = d_lr + (size_t)b * lr_capacity;
    for (int i = tid; i < nlr; i += 256) {
        const int q = lr[i].queryIdx;
        if (q >= 0 && q < kp_capacity) kp2lr[q] = i;
    }
    __syncthreads();
    const int nm = min(d_nm[b], match_capacity);
    const vslam_dmatch* m = d_m + (size_t)b * match_capacity;
    int written = 0;
    for (int base = 0; base < nm; base += 256) {
        const int i = base + tid;
        bool ok = false; int li = -1, ti = 0;
        if (i < nm) {
            const int q = m[i].queryIdx; ti = m[i].trainIdx;
            if (q >= 0 && q < kp_capacity && ti >= 0 && ti < kp_capacity) { li = kp2lr[q]; ok = li >= 0 && d_valid_lr[(size_t)b * lr_capacity + li] != 0; }
        }
        const unsigned long long mask = __ballot(ok);
        __syncthreads();
        if (lane == 0) s_tot[wave] = __popcll(mask);
        __syncthreads();
        int off = written;
        for (int w = 0; w < wave; ++w) off += s_tot[w];
        const int slot = off + __popcll(mask & ((1ull << lane) - 1ull));
        if (ok && slot < out_capacity) {
            const float* p = d_xyz_lr + 3 * ((size_t)b * lr_capacity + li);
            float* o = d_xyz_out + 3 * ((size_t)b * out_capacity + slot);
            o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
            const vslam_keypoint* k = d_kpsT + (size_t)b * kp_capacity + ti;
            reinterpret_cast<float2*>(d_uv_out)[(size_t)b * out_capacity + slot] = make_float2(k->x, k->y);
        }
        written += s_tot[0] + s_tot[1] + s_tot[2] + s_tot[3];
    }
    if (tid == 0) d_nout[b] = min(written, out_capacity);
}

int launch_build_pnp_inputs(const vslam_dmatch* d_m, const int32_t* d_nm, int match_capacity, const vslam_dmatch* d_lr,
                            const int32_t* d_nlr, int lr_capacity, const float* d_xyz_lr, const uint8_t* d_valid_lr,
                            const vslam_keypoint* d_kpsT, int kp_capacity, int B, int32_t* d_kp2lr, float* d_xyz_out, float* d_uv_out,
                            int32_t* d_nout, int out_capacity, hipStream_t stream) {
    if (B <= 0) return VSLAM_OK;
    ProfScope prof__(stream, "build_pnp_inputs_kernel");
    hipLaunchKernelGGL(build_pnp_inputs_kernel, dim3(B), dim3(256), 0, stream, d_m, d_nm, match_capacity, d_lr, d_nlr, lr_capacity, d_xyz_lr,
                       d_valid_lr, d_kpsT, kp_capacity, d_kp2lr, d_xyz_out, d_uv_out, d_nout, out_capacity);
    VS_HIP(hipGetLastError());
    return VSLAM_OK;
}

int launch_find3d_disparity(const vslam_keypoint* d_kps, int n, const float* d_disp, int w, int h, int dstride, const double* d_T,
                            CamParams cam, float* d_xyz, uint8_t* d_valid, uint8_t* d_rel, hipStream_t stream) {
    if (n <= 0) return VSLAM_OK;
    hipLaunchKernelGGL(find3d_disparity_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, d_kps, n, d_disp, w, h, dstride, d_T, cam,
                       d_xyz, d_valid, d_rel);
    VS_HIP(hipGetLastError());
    return VSLAM_OK;
}

int launch_find3d_disparity_batch(const vslam_keypoint* d_kps, const int32_t* d_n, int kp_capacity, int B, const float* d_disp, int w, int h,
                                  const double* d_T, CamParams cam, float* d_xyz, uint8_t* d_valid, uint8_t* d_rel, hipStream_t stream) {
    if (B <= 0) return VSLAM_OK;
    ProfScope prof__(stream, "find3d_disparity_kernel");
    hipLaunchKernelGGL(find3d_disparity_batch_kernel, dim3((kp_capacity + 255) / 256, B), dim3(256), 0, stream, d_kps, d_n, kp_capacity, d_disp, w, h,
                       d_T, cam, d_xyz, d_valid, d_rel);
    VS_HIP(hipGetLastError());
    return VSLAM_OK;
}

int launch_triangulate(const float* d_uvL, const float* d_uvR, const int32_t* d_n, int capacity, int B, const double* d_T, CamParams cam,
                       float* d_xyz, uint8_t* d_valid, uint8_t* d_rel, hipStream_t stream) {
    if (B <= 0 || capacity <= 0) return VSLAM_OK;
    ProfScope prof__(stream, "triangulate_kernel");
    hipLaunchKernelGGL(triangulate_kernel, dim3((capacity + 255) / 256, B), dim3(256), 0, stream, d_uvL, d_uvR, d_n, capacity, d_T, cam,
                       d_xyz, d_valid, d_rel);
    VS_HIP(hipGetLastError());
    return VSLAM_OK;
}

int launch_gather_uv(const vslam_keypoint* d_kpsQ, const vslam_keypoint* d_kpsT, int kp_capacity, const vslam_dmatch* d_m,
                     const int32_t* d_nm, int match_capacity, int B, float* d_uvQ, float* d_uvT, hipStream_t stream) {
    if (B <= 0 || match_capacity <= 0) return VSLAM_OK;
    ProfScope prof__(stream, "gather_uv_kernel");
    hipLaunchKernelGGL(gather_uv_kernel, dim3((match_capacity + 255) / 256, B), dim3(256), 0, stream, d_kpsQ, d_kpsT, kp_capacity, d_m,
                       d_nm, match_capacity, d_uvQ, d_uvT);
    VS_HIP(hipGetLastError());
    return VSLAM_OK;
}

// ---- measurement aid: what a plain streaming copy reaches on this box (the achievable HBM ceiling next to the 8 TB/s spec).
// Several shapes of the same 16-B-per-lane copy; bench.py reports the best one and its name.  A workgroup owns contiguous
// 256 x U x 16 B chunks (grid-stride over chunks), issues its U independent loads before the first store; NT = non-temporal
// loads and stores (the copy has no reuse, so it should not displace L2 / Infinity Cache lines).
typedef float v4f __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void hbm_copy_probe_kernel(const v4f* __restrict__ src, v4f* __restrict__ dst, size_t n) {
    const size_t chunk = (size_t)256 * U, nchunks = n / chunk;
    for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const size_t base = c * chunk + threadIdx.x;
        v4f r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) r[u] = NT ? __builtin_nontemporal_load(src + base + (size_t)u * 256) : src[base + (size_t)u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (NT) __builtin_nontemporal_store(r[u], dst + base + (size_t)u * 256);
            else dst[base + (size_t)u * 256] = r[u];
        }
    }
    if (blockIdx.x == 0)
        for (size_t i = nchunks * chunk + threadIdx.x; i < n; i += 256) dst[i] = src[i];
}
struct CopyVariant { const char* name; int U; bool nt; int wg_per_cu; };
static const CopyVariant kCopyVariants[] = {
    {"U4 plain, 8 WG/CU", 4, false, 8},        {"U8 plain, 8 WG/CU", 8, false, 8},   {"U8 nontemporal, 8 WG/CU", 8, true, 8},
    {"U8 nontemporal, 16 WG/CU", 8, true, 16}, {"U8 nontemporal, 32 WG/CU", 8, true, 32}, {"U16 nontemporal, 8 WG/CU", 16, true, 8},
    {"U4 nontemporal, one chunk per WG", 4, true, 0}, {"U8 plain, one chunk per WG", 8, false, 0},
    {"U2 nontemporal, one chunk per WG", 2, true, 0}, {"U1 plain, one float4 per thread", 1, false, 0}, {"U1 nontemporal, one float4 per thread", 1, true, 0},
    {"U2 plain, one chunk per WG", 2, false, 0},
};
int hbm_copy_probe_variants() { return (int)(sizeof(kCopyVariants) / sizeof(kCopyVariants[0])); }
const char* hbm_copy_probe_name(int v) { return (v >= 0 && v < hbm_copy_probe_variants()) ? kCopyVariants[v].name : ""; }
int launch_hbm_copy_probe(const void* src, void* dst, size_t bytes, int variant, hipStream_t stream) {
    if (variant < 0 || variant >= hbm_copy_probe_variants()) return VSLAM_ERR_ARG;
    const CopyVariant& cv = kCopyVariants[variant];
    const size_t n = bytes / 16;
    const size_t nchunks = n / ((size_t)256 * cv.U);
    size_t grid = cv.wg_per_cu > 0 ? (size_t)256 * cv.wg_per_cu : nchunks;
    if (grid > nchunks) grid = nchunks;
    if (grid < 1) grid = 1;
    const v4f* s = (const v4f*)src; v4f* d = (v4f*)dst;
#define VS_COPY(UU, NTT) hipLaunchKernelGGL((hbm_copy_probe_kernel<UU, NTT>), dim3((unsigned)grid), dim3(256), 0, stream, s, d, n)
    if (cv.U == 1 && !cv.nt) VS_COPY(1, false);
    else if (cv.U == 1) VS_COPY(1, true);
    else if (cv.U == 2 && !cv.nt) VS_COPY(2, false);
    else if (cv.U == 2) VS_COPY(2, true);
    else if (cv.U == 4 && !cv.nt) VS_COPY(4, false);
    else if (cv.U == 4) VS_COPY(4, true);
    else if (cv.U == 8 && !cv.nt) VS_COPY(8, false);
    else if (cv.U == 8) VS_COPY(8, true);
    else VS_COPY(16, true);
#undef VS_COPY
    VS_HIP(hipGetLastError());
    return VSLAM_OK;
}

} // namespace vslam
