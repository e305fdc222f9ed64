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
// gfx950 mapping.  The all-pairs Hamming table is the one dense contraction of the pipeline: with the descriptor bits
// recoded as +-1 bytes, <a, b> = 256 - 2 hamming(a, b), so a 32 x 32 block of distances is eight
// v_mfma_i32_32x32x32_i8 (exact integer arithmetic).  The +-1 form never exists in HBM: `match_train_nearest_kernel` reads the raw
// 32-B descriptors and expands bits to bytes in registers -- the 64 train columns of a wave once, into the VGPRs that stay resident
// as MFMA B operands; every 32-row query tile cooperatively (one dword -> 32 bytes per thread) on its way into LDS (row stride
// 272 B: conflict-free ds_read_b128), shared by the four waves of the workgroup -- and folds the column minima in the
// accumulator layout (column = lane % 32): per value one v_lshl_add + one v_max on packed keys, with the +256 bias
// supplied as the MFMA C operand.  The previous v_xor + v_bcnt formulation was bound by v_bcnt_u32_b32 issuing at quarter rate.
#include "vslam_internal.h"

namespace vslam {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// four descriptor bits -> four bytes of +-1 (bit set -> 0xFF = -1, clear -> 0x01 = +1); bit k of the nibble -> byte k
__device__ inline uint32_t expand_nibble(uint32_t nib) {
    const uint32_t s = __umul24(nib & 0xFu, 0x204081u) & 0x01010101u;
    return ((s << 8) - (s << 1)) | 0x01010101u;
}
// 16 descriptor bits -> 16 operand bytes
__device__ inline int4 expand_half(uint32_t bits16) {
    return make_int4((int)expand_nibble(bits16), (int)expand_nibble(bits16 >> 4), (int)expand_nibble(bits16 >> 8), (int)expand_nibble(bits16 >> 12));
}

#ifndef VSLAM_MATCH_CT
#define VSLAM_MATCH_CT 4
#endif
constexpr int kMatchBlock = 256;             // 4 waves
constexpr int kColTiles = VSLAM_MATCH_CT;    // 32-column MFMA tiles per wave held in registers (2: accumulators double-buffered; 4: one A read feeds four tiles -- 0.137 vs 0.148 ms per 256 items of 1500 x 1500, 0.103 vs 0.099 at 1100, 0.046 vs 0.044 at 700)
constexpr int kColsPerWave = 32 * kColTiles;
constexpr int kColsPerBlock = (kMatchBlock / 64) * kColsPerWave;
constexpr int kQRows = 32;                   // query rows per LDS tile
constexpr int kQStride = 272;                // bytes per staged query row (256 + 16: conflict-free 16-B reads)

// epilogue of one 32-row query tile: fold the 16 accumulator slots of this lane (per column tile) into the running maximum of
// dot << 16 | (0xFFFF - row) (signed compare: largest dot = smallest distance, then smallest row)
__device__ inline void fold_tile(const v16i (&acc)[kColTiles], int ib, int nq, bool full, int (&m)[kColTiles]) {
    const int inv = 0xFFFF - ib;
    if (full) {
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int iv = inv - (8 * (v / 4) + (v % 4));
#pragma unroll
            for (int t = 0; t < kColTiles; ++t) m[t] = max(m[t], (acc[t][v] << 16) + iv);
        }
    } else { // last, partial tile: rows >= nq must not compete
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int off = 8 * (v / 4) + (v % 4);
            if (ib + off < nq) {
#pragma unroll
                for (int t = 0; t < kColTiles; ++t) m[t] = max(m[t], (acc[t][v] << 16) + inv - off);
            }
        }
    }
}

// Work items are numbered column-block-major (w = (colblock * qsplit + split) * B + item): the items' live column blocks
// come first in the grid and spread evenly over the XCDs / CUs, the blocks beyond an item's train rows (capacity padding)
// sit at the end and exit at once.  (With the item as the slow index the live blocks of every item landed on the same few XCDs.)
__global__ __launch_bounds__(kMatchBlock) void match_train_nearest_kernel(
    const uint8_t* __restrict__ d_q, size_t q_stride, const int32_t* __restrict__ d_nq, const uint8_t* __restrict__ d_t, size_t t_stride,
    const int32_t* __restrict__ d_nt, int max_rows, int qsplit, int B, uint32_t* __restrict__ d_train_best) {
    const int w = blockIdx.x;
    const int b = w % B, split = (w / B) % qsplit, cb = w / (B * qsplit);
    const int nq = min(d_nq[b], max_rows), nt = min(d_nt[b], max_rows);
    const int c0 = cb * kColsPerBlock;
    if (c0 >= nt || nq <= 0) return;
    // query range of this split, in whole tiles
    const int ntiles = (nq + kQRows - 1) / kQRows, per = (ntiles + qsplit - 1) / qsplit;
    const int tile0 = split * per, tile1 = min(ntiles, tile0 + per);
    if (tile0 >= tile1) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    __shared__ alignas(16) int8_t sq[2][kQRows * kQStride];
    __shared__ uint2 lut[256]; // descriptor byte -> its eight +-1 operand bytes
    {
        const uint32_t v = threadIdx.x;
        lut[v] = make_uint2(expand_nibble(v), expand_nibble(v >> 4));
    }
    const uint8_t* Q = d_q + (size_t)b * q_stride;
    const uint8_t* T = d_t + (size_t)b * t_stride;
    // B operands: this wave's 64 train columns, expanded once and resident for the whole kernel.  Operand slice s of lane
    // (r, h) holds k = 32 s + 16 h .. + 15, i.e. halfword 2 s + h of the descriptor.
    v4i breg[kColTiles][8];
#pragma unroll
    for (int t = 0; t < kColTiles; ++t) {
        const int j = min(c0 + wave * kColsPerWave + 32 * t + r, nt - 1);
        const uint4 lo = *reinterpret_cast<const uint4*>(T + (size_t)j * 32), hi = *reinterpret_cast<const uint4*>(T + (size_t)j * 32 + 16);
        const uint32_t wd[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int4 e = expand_half((wd[s] >> (16 * h)) & 0xFFFFu);
            breg[t][s] = v4i{e.x, e.y, e.z, e.w};
        }
    }
    // staging: a tile is 32 rows x 8 dwords of raw descriptor = one dword per thread, expanded to 32 operand bytes into LDS
    const int srow = threadIdx.x >> 3, sword = threadIdx.x & 7;
    auto gload = [&](int tile, uint32_t& x) {
        const int r0 = min(tile * kQRows + srow, nq - 1);
        x = reinterpret_cast<const uint32_t*>(Q + (size_t)r0 * 32)[sword];
    };
    auto sstore = [&](int buf, uint32_t x) {
        const uint2 b0 = lut[x & 0xFFu], b1 = lut[(x >> 8) & 0xFFu], b2 = lut[(x >> 16) & 0xFFu], b3 = lut[x >> 24];
        int4* dst = reinterpret_cast<int4*>(&sq[buf][srow * kQStride + sword * 32]);
        dst[0] = make_int4((int)b0.x, (int)b0.y, (int)b1.x, (int)b1.y); dst[1] = make_int4((int)b2.x, (int)b2.y, (int)b3.x, (int)b3.y);
    };
    auto mma_tile = [&](int buf, v16i (&acc)[kColTiles]) {
        v4i a[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) a[s] = *reinterpret_cast<const v4i*>(&sq[buf][r * kQStride + s * 32 + h * 16]);
        const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; // (the first MFMA of a tile reads C = 0 as an inline constant: no accumulator clears)
#pragma unroll
        for (int t = 0; t < kColTiles; ++t) acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[0], breg[t][0], zero, 0, 0, 0);
#pragma unroll
        for (int s = 1; s < 8; ++s)
#pragma unroll
            for (int t = 0; t < kColTiles; ++t) acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s], breg[t][s], acc[t], 0, 0, 0);
    };
    uint32_t x;
    gload(tile0, x);
    __syncthreads(); // the expansion table is complete
    sstore(0, x);
    if (tile0 + 1 < tile1) gload(tile0 + 1, x);
    __syncthreads();
    int m[kColTiles];
#pragma unroll
    for (int t = 0; t < kColTiles; ++t) m[t] = INT_MIN;
    if constexpr (kColTiles <= 2) {
        // software pipeline: the MFMAs of tile t + 1 are issued before the (VALU) epilogue of tile t (two accumulator sets)
        v16i accA[kColTiles], accB[kColTiles];
        mma_tile(0, accA);
        for (int tile = tile0; tile < tile1; tile += 2) {
            // ---- stage tile + 1 into buffer 1, issue its MFMAs, then fold tile
            if (tile + 1 < tile1) { sstore(1, x); if (tile + 2 < tile1) gload(tile + 2, x); }
            __syncthreads();
            if (tile + 1 < tile1) mma_tile(1, accB);
            fold_tile(accA, tile * kQRows + 4 * h, nq, tile * kQRows + kQRows <= nq, m);
            if (tile + 1 >= tile1) break;
            // ---- stage tile + 2 into buffer 0, issue its MFMAs, then fold tile + 1
            if (tile + 2 < tile1) { sstore(0, x); if (tile + 3 < tile1) gload(tile + 3, x); }
            __syncthreads();
            if (tile + 2 < tile1) mma_tile(0, accA);
            fold_tile(accB, (tile + 1) * kQRows + 4 * h, nq, (tile + 1) * kQRows + kQRows <= nq, m);
        }
    } else {
        // four column tiles per wave: one set of accumulators; a query tile read from LDS once feeds 32 MFMAs (half the LDS
        // traffic per MAC); the other wave of the SIMD fills the MFMA pipe while this one folds
        v16i acc[kColTiles];
        for (int tile = tile0; tile < tile1; ++tile) {
            const int buf = (tile - tile0) & 1;
            if (tile + 1 < tile1) { sstore(buf ^ 1, x); if (tile + 2 < tile1) gload(tile + 2, x); } // (buffer buf ^ 1 was released by the barrier below)
            mma_tile(buf, acc);
            fold_tile(acc, tile * kQRows + 4 * h, nq, tile * kQRows + kQRows <= nq, m);
            __syncthreads();
        }
    }
#pragma unroll
    for (int t = 0; t < kColTiles; ++t) m[t] = max(m[t], __shfl_xor(m[t], 32));
    if (h == 0) {
#pragma unroll
        for (int t = 0; t < kColTiles; ++t) {
            const int mm = m[t];
            const int j = c0 + wave * kColsPerWave + 32 * t + r;
            if (j < nt && mm != INT_MIN) {
                const uint32_t d = (uint32_t)(256 - (mm >> 16)) >> 1, i = 0xFFFFu - ((uint32_t)mm & 0xFFFFu);
                const uint32_t key = (d << 16) | i;
                if (qsplit == 1) d_train_best[(size_t)b * max_rows + j] = key;
                else atomicMin(&d_train_best[(size_t)b * max_rows + j], key);
            }
        }
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
    const int tblocks = (max_rows + kColsPerBlock - 1) / kColsPerBlock;
    int qsplit = 1;
    while (qsplit < 16 && (long)tblocks * qsplit * B < 1024 && max_rows / (qsplit * 2) >= 4 * kQRows) qsplit *= 2;
    // every in-range train row is written exactly once when the query range is not split
    if (qsplit > 1) VS_HIP(hipMemsetAsync(d_train_best, 0xFF, (size_t)B * max_rows * sizeof(uint32_t), stream));
    {
        ProfScope prof__(stream, "match_train_nearest_kernel");
        hipLaunchKernelGGL(match_train_nearest_kernel, dim3(tblocks * qsplit * B), dim3(kMatchBlock), 0, stream, d_q, q_stride, d_nq, d_t, t_stride,
                           d_nt, max_rows, qsplit, B, d_train_best);
    }
    ProfScope prof__(stream, "match_finalize_kernel");
    hipLaunchKernelGGL(match_finalize_kernel, dim3(B), dim3(kFinBlock), 0, stream, d_nq, d_nt, d_gap, gate, ratio, gap_thr,
                       max_rows, d_train_best, d_out, out_capacity, d_nout);
    VS_HIP(hipGetLastError());
    return VSLAM_OK;
}

} // namespace vslam
