// match_kernels.hip -- K7: brute-force cross-checked Hamming matcher + distance gate.
//
// Replaces cv::BFMatcher(NORM_HAMMING, crossCheck=true)::match + the gate of VO::feature_matching
// (/root/reference/src/stereo_visual_slam_main/visual_odometry.cpp:219-251).  Semantics (OpenCV 3.2
// batchDistance with crosscheck): (i) every TRAIN row j takes its nearest QUERY row i*(j), first minimum in
// ascending i; (ii) for j ascending, query i*(j) takes train j if d < dist[i*].  Both steps are order-free
// minima of packed keys, so they map to atomicMin without changing the result:
//   step (i)  : key = d << 16 | i   -> min over i  = smallest d, then smallest i   (first minimum)
//   step (ii) : key = d << 16 | j   -> min over {j : i*(j) = i} = smallest d, then smallest j (strict '<').
//
// gfx950 mapping: one lane owns one train descriptor (8 dwords in VGPRs); query descriptors are staged through
// LDS in 16-B slots and broadcast-read (all lanes read the same address: conflict-free); distance =
// 8 x (v_xor_b32 + v_bcnt_u32_b32 accumulate).  Integer work only: HBM/LDS-issue bound, no MFMA.
#include "vslam_internal.h"

namespace vslam {

constexpr int kMatchBlock = 256;
constexpr int kQTile = 128; // queries staged per LDS tile (4 KiB)

__global__ __launch_bounds__(kMatchBlock) void match_train_nearest_kernel(
    const uint8_t* __restrict__ d_q, size_t q_stride, const int32_t* __restrict__ d_nq, const uint8_t* __restrict__ d_t,
    size_t t_stride, const int32_t* __restrict__ d_nt, int max_rows, int qsplit, uint32_t* __restrict__ d_train_best) {
    const int b = blockIdx.z;
    const int nq = min(d_nq[b], max_rows), nt = min(d_nt[b], max_rows);
    const int j = blockIdx.x * kMatchBlock + threadIdx.x;
    if (blockIdx.x * kMatchBlock >= nt || nq <= 0) return;
    // query range of this split
    const int per = (nq + qsplit - 1) / qsplit;
    const int q0 = blockIdx.y * per, q1 = min(nq, q0 + per);
    if (q0 >= q1) return;

    __shared__ uint4 sq[kQTile * 2];
    const uint4* tq = reinterpret_cast<const uint4*>(d_t + (size_t)b * t_stride);
    uint4 t0 = make_uint4(0, 0, 0, 0), t1 = t0;
    if (j < nt) { t0 = tq[2 * j]; t1 = tq[2 * j + 1]; }
    const uint4* gq = reinterpret_cast<const uint4*>(d_q + (size_t)b * q_stride);

    uint32_t best = 0xFFFFFFFFu;
    for (int base = q0; base < q1; base += kQTile) {
        const int cnt = min(kQTile, q1 - base);
        __syncthreads();
        for (int s = threadIdx.x; s < cnt * 2; s += kMatchBlock) sq[s] = gq[2 * base + s];
        __syncthreads();
        for (int i = 0; i < cnt; ++i) {
            const uint4 a = sq[2 * i], c = sq[2 * i + 1];
            uint32_t d = __popc(a.x ^ t0.x);
            d += __popc(a.y ^ t0.y);
            d += __popc(a.z ^ t0.z);
            d += __popc(a.w ^ t0.w);
            d += __popc(c.x ^ t1.x);
            d += __popc(c.y ^ t1.y);
            d += __popc(c.z ^ t1.z);
            d += __popc(c.w ^ t1.w);
            const uint32_t key = (d << 16) | (uint32_t)(base + i);
            best = min(best, key);
        }
    }
    if (j < nt) {
        if (qsplit == 1) d_train_best[(size_t)b * max_rows + j] = best;
        else atomicMin(&d_train_best[(size_t)b * max_rows + j], best);
    }
}

// ordered compaction helper: returns the exclusive rank of `flag` among the block's threads, and the block total
__device__ inline int block_rank(bool flag, int* s_wave_tot, int& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const unsigned long long m = __ballot(flag);
    const int rank = __popcll(m & ((1ull << lane) - 1ull));
    __syncthreads();
    if (lane == 0) s_wave_tot[wave] = __popcll(m);
    __syncthreads();
    int base = 0, tot = 0;
    for (int w = 0; w < nwaves; ++w) {
        const int c = s_wave_tot[w];
        if (w < wave) base += c;
        tot += c;
    }
    total = tot;
    return base + rank;
}

constexpr int kFinBlock = 1024;

__global__ __launch_bounds__(kFinBlock) void match_finalize_kernel(
    const int32_t* __restrict__ d_nq, const int32_t* __restrict__ d_nt, const double* __restrict__ d_gap, int gate,
    double ratio, double gap_thr, int max_rows, const uint32_t* __restrict__ d_train_best, vslam_dmatch* __restrict__ d_out,
    int out_capacity, int32_t* __restrict__ d_nout) {
    const int b = blockIdx.x;
    const int nq = min(d_nq[b], max_rows), nt = min(d_nt[b], max_rows);
    __shared__ uint32_t qbest[kMaxRows];
    __shared__ int s_wave_tot[kFinBlock / 64];
    __shared__ uint32_t s_min[kFinBlock / 64];
    for (int i = threadIdx.x; i < kMaxRows; i += kFinBlock) qbest[i] = 0xFFFFFFFFu;
    __syncthreads();
    if (nq > 0)
        for (int j = threadIdx.x; j < nt; j += kFinBlock) {
            const uint32_t key = d_train_best[(size_t)b * max_rows + j];
            if (key != 0xFFFFFFFFu) atomicMin(&qbest[key & 0xFFFFu], (key & 0xFFFF0000u) | (uint32_t)j);
        }
    __syncthreads();
    // d_min over the matches (visual_odometry.cpp:229-234)
    uint32_t dmin = 0xFFFFu;
    for (int i = threadIdx.x; i < nq; i += kFinBlock) {
        const uint32_t key = qbest[i];
        if (key != 0xFFFFFFFFu) dmin = min(dmin, key >> 16);
    }
    for (int o = 32; o > 0; o >>= 1) dmin = min(dmin, (uint32_t)__shfl_xor((int)dmin, o));
    if ((threadIdx.x & 63) == 0) s_min[threadIdx.x >> 6] = dmin;
    __syncthreads();
    dmin = 0xFFFFu;
    for (int w = 0; w < kFinBlock / 64; ++w) dmin = min(dmin, s_min[w]);
    // threshold (visual_odometry.cpp:242), evaluated in double like the reference
    const double a = ratio * (double)(float)dmin, g = gap_thr * d_gap[b];
    const double thr = a > g ? a : g;
    vslam_dmatch* out = d_out + (size_t)b * out_capacity;
    int written = 0;
    for (int base = 0; base < nq; base += kFinBlock) {
        const int i = base + threadIdx.x;
        uint32_t key = 0xFFFFFFFFu;
        if (i < nq) key = qbest[i];
        bool keep = key != 0xFFFFFFFFu;
        const uint32_t d = key >> 16;
        if (keep && gate) keep = (double)(float)d <= thr;
        int total;
        const int r = block_rank(keep, s_wave_tot, total);
        if (keep && written + r < out_capacity) {
            vslam_dmatch m;
            m.queryIdx = i; m.trainIdx = (int)(key & 0xFFFFu); m.imgIdx = 0; m.distance = (float)d;
            out[written + r] = m;
        }
        written += total;
    }
    if (threadIdx.x == 0) d_nout[b] = min(written, out_capacity);
}

int launch_match(const uint8_t* d_q, size_t q_stride, const int32_t* d_nq, const uint8_t* d_t, size_t t_stride,
                 const int32_t* d_nt, const double* d_gap, int gate, double ratio, double gap_thr, int B, int max_rows,
                 uint32_t* d_train_best, vslam_dmatch* d_out, int out_capacity, int32_t* d_nout, hipStream_t stream) {
    if (B <= 0) return VSLAM_OK;
    if (max_rows > kMaxRows || max_rows <= 0) { set_error("matcher: max_rows %d out of range (<= %d)", max_rows, kMaxRows); return VSLAM_ERR_ARG; }
    // fill the chip: ~>= 1024 workgroups.  Split the query range when the batch is small.
    const int tblocks = (max_rows + kMatchBlock - 1) / kMatchBlock;
    int qsplit = 1;
    while (qsplit < 16 && (long)tblocks * qsplit * B < 1024 && max_rows / (qsplit * 2) >= kQTile) qsplit *= 2;
    if (qsplit > 1) VS_HIP(hipMemsetAsync(d_train_best, 0xFF, (size_t)B * max_rows * sizeof(uint32_t), stream));
    {
        ProfScope prof__(stream, "match_train_nearest_kernel");
        hipLaunchKernelGGL(match_train_nearest_kernel, dim3(tblocks, qsplit, B), dim3(kMatchBlock), 0, stream, d_q, q_stride, d_nq,
                           d_t, t_stride, d_nt, max_rows, qsplit, d_train_best);
    }
    ProfScope prof__(stream, "match_finalize_kernel");
    hipLaunchKernelGGL(match_finalize_kernel, dim3(B), dim3(kFinBlock), 0, stream, d_nq, d_nt, d_gap, gate, ratio, gap_thr,
                       max_rows, d_train_best, d_out, out_capacity, d_nout);
    VS_HIP(hipGetLastError());
    return VSLAM_OK;
}

} // namespace vslam
