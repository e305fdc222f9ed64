// ba_resident.hip -- optimize_map (/root/reference/src/stereo_visual_slam_main/optimization.cpp:103-288) and the per-keyframe schedule of
// run_vslam.cpp:58-66 on windows whose landmark state FITS THE LDS OF ONE CU.
//
// lm_kernels.hip keeps a window's working set (positions, per-landmark blocks, per-edge weights, Schur hit lists: ~1.3 MB) in HBM and
// walks it with one dependent round trip per phase; at two waves per SIMD nothing hides those trips, and on the windows a real sequence
// produces (3.5 k landmarks, 4.6 k edges) an LM iteration took ~280 k cycles for ~40 k cycles of arithmetic.  This kernel turns the
// design around:
//   * the landmark positions (f64, 24 B each) live in LDS for the whole schedule -- [r5] those of every multi-observation row and of as many
//     single-observation rows as half a CU's LDS holds (two windows share a CU, see kRsBlock); the other singles rows, visited once per pass
//     and in order, keep their working positions in a global array requested a row ahead; the other per-iteration global traffic is the
//     read-only observation table (8 B per edge, slot-major, coalesced, L2-resident across iterations) and one fire-and-forget backup of the
//     accepted positions per accepted step.  Nothing else is stored: no H_ll, b_l, D^-1, Huber weights, keyframe-major copies, hit lists.
//   * landmarks are SORTED by (observation count descending, keyframe set): a row of 64 consecutive landmarks then sees (almost always)
//     one keyframe set, so "which keyframe is observation q" is wave-uniform and ONE landmark-wise pass does everything the old kernel
//     spread over four phases: it evaluates the landmark's observations, forms H_ll, b_l, D^-1 in registers and, for every pair of its
//     keyframes, the 6x6 block  At1^T M At2  of the Schur complement, which a halving butterfly sums over the row.
//   * the row sums go into the reduced system with INTEGER atomics on a common fixed-point scale (2^60 / bound, the bound summed
//     per observation by the pass that evaluated the state): integer addition is associative, so rows can be handed to waves dynamically
//     and the result is still bit-reproducible -- the property the adaptive schedule's "continue instead of repeat" argument needs.
//   * the back-substitution re-derives D^-1 and b_l from the landmark's own observations, updates the position IN PLACE in LDS and
//     evaluates the trial cost in the same visit; a rejected step restores the positions from the backup (rare).
// Reduced system: blocked right-looking Cholesky as in lm_kernels.hip, on a block-packed lower triangle (nk (nk + 1) / 2 blocks of 36).
// Windows that do not fit (multi-observation rows x 1536 B + 2 B per landmark + blocks beyond the LDS budget, > 5120 landmarks) are marked deferred and taken by
// lm_window_kernel in the same launch set.  Same LM rules (g2o Levenberg, Huber 5.991, <= 10 trials), same chi2 classification, same
// statistics as lm_kernels.hip; sums differ from it in rounding only.
#include "vslam_internal.h"

#include "lm_device.h"

namespace vslam {

#ifndef VSLAM_RS_MIN_WAVES
#define VSLAM_RS_MIN_WAVES 2 // waves per SIMD the register allocation must leave room for (2: 256 VGPRs)
#endif
// [r5, second session] 256 lanes per window, TWO windows per CU (the launcher asks for half a CU's LDS): with 512 lanes a window held a whole CU
// -- all of its registers, 156 KB of LDS -- at 20-27 % VALU activity, and its serial stretches (the 60-pivot factorisation: one or two waves busy
// for 23 % of the time, barriers, exposed round trips) idled the other waves.  Two windows of four waves each use the same registers and fill each
// other's gaps: 2.27 -> 1.82 ms per 512 small windows in the experiment that led here (profiles/r05_experiments.log).  What made a window fit half
// the LDS: the positions of the SINGLES rows (87 % of the landmarks) live in global memory (L2), see `rl0` below.
// Two widths of the same kernel (template parameter BLOCK): 256 lanes and half a CU's LDS per window -- two windows per CU, the throughput form,
// used when a launch has more windows than the device has CUs -- and 512 lanes with a whole CU's LDS for small launches, where a window's LATENCY
// is what counts (a 50-frame sequence pass, a host-tier call).  Both give the same bits: every floating-point sum that crosses waves is
// taken over EIGHT row streams (row r belongs to stream r mod 8; a wave of the narrow form carries two of them), everything else that crosses
// waves is an integer sum or a maximum.  tests/test_gpu_ba_resident.py::test_both_widths_bit_identical.
constexpr int kRsStreams = 8;
constexpr int kRsWavesMax = 8;
constexpr int kRsKf = VSLAM_MAX_KF;
constexpr int kRsNp = 6 * kRsKf;
constexpr int kRsMaxRowsMax = 20 * kRsWavesMax; // 64-landmark rows per window: 20 per wave (the classification keeps six bits per row slot in two 64-bit words per lane): 5120 / 10240 landmarks
constexpr int kRsSlotsReg = 6;                 // observations per landmark preloaded into registers (the rest is fetched where it is used)
constexpr int kRsSchedFinalIters = 10;         // run_vslam.cpp:66
constexpr int kRsSortCap = 8192;               // keys of the in-kernel landmark sort (power of two >= landmarks)
constexpr int kRsDbg = 16;
constexpr int kRsPairs = kRsKf * (kRsKf + 1) / 2;
constexpr int kRsHitChunk = 4;                 // 64-hit rows per work item of the hit-major Schur phase
constexpr int kRsHitItems = 512;
constexpr int kRsChunk = 4;                    // singles rows per work item of the linearisation pass
constexpr int kRsHitReserveMax = 24 * 1024;    // most LDS bytes kept free for the hit lists when rows are dealt between LDS and global memory

struct alignas(16) RsShared {
    double T[kRsKf * 7], TT[kRsKf * 7];
    double Rt[kRsKf * 12], RtT[kRsKf * 12];
    double bp[kRsNp], bs[kRsNp], xp[kRsNp], rdiag[kRsNp];
    double Ld[kRsKf * 24];
    long long bpq[kRsNp], bsq[kRsNp], hdq[kRsNp]; // fixed-point accumulators: pose gradient, reduced right-hand side, diag(H_pp)
    double red[kRsStreams * 2];
    int redi[kRsWavesMax * 8];
    int slotoff[kRsKf + 2];                       // observation slot q of sorted landmark s sits at entry slotoff[q] + s of the slot-major tables (may be negative: slot q starts at landmark nl - n_q)
    int nq[kRsKf + 2];
    int flag[16];
    unsigned short rowU[kRsMaxRowsMax];           // union of the keyframe sets of a row's live landmarks
    unsigned char rowC[kRsMaxRowsMax];            // most observations of a live landmark in the row
    int pairoff[kRsPairs + 2];                    // first hit of keyframe pair p (k1 <= k2, diagonal pairs included) in the LDS hit list
    int itemoff[kRsPairs + 2];                    // first work item (chunk of kRsHitChunk hit rows) of pair p
    unsigned char pk1[kRsPairs + 2], pk2[kRsPairs + 2];
    unsigned char itempair[kRsHitItems];           // pair of work item i of the hit-major phase
};

struct RsArgs {
    LmWindowArgs a;
    float2* uv_s;          // total_edge: observations, slot-major per window: slot q of sorted landmark s at [slotoff[q] + s]
    int32_t* epos;         // total_edge: caller's edge id (window-local) at the same position
    double* tab;           // 6 x total_lm doubles of scratch; window slice [6 lm0, 6 (lm0 + nl)) holds, as u16: perm[nl] (sorted position ->
                           // landmark id, window-local), mstat[nl] (keyframe set (12 bits) | slot of the landmark's last edge << 12, sorted order)
    double* xin;           // 3 x total_lm doubles of scratch; window slice holds, as f32, the input positions in sorted order (3 nl floats)
    double* Pbak;          // 3 x total_lm: the accepted positions while a trial sits in LDS
    double* Dc;            // 6 x total_lm: D^-1 = (H_ll + lambda I)^-1 of the multi-observation landmarks, sorted order, 48 B records (hit-major phase)
    double* blc;           // 3 x total_lm: their b_l, 24 B records
    int32_t* status;
    int32_t* passes;
    int32_t* defer;        // n_windows: 1 = window left to lm_window_kernel
    const int32_t* order;
    long long* dbg;
    int dyn_bytes;
    int dyn_narrow;        // the dynamic LDS of the 256-lane form: WHICH windows the kernel takes and on which path (hit-major / row-wise) is decided against it in both widths, so that a window's bits do not depend on the width
    int want_chi2;
    int dense_to_general;  // 1: windows with more than 2.2 observations per landmark are left to lm_window_kernel (Tuning::ba_resident = 1 forces them here)
};

__device__ inline long long to_fixed(double v, double scale) { return __double2ll_rn(v * scale); }

// lower block (I >= K) of the packed reduced system
__device__ inline int rs_blk(int I, int K) { return (I * (I + 1) / 2 + K) * 36; }

template <bool SCHED, int BLOCK>
__global__ __launch_bounds__(BLOCK, VSLAM_RS_MIN_WAVES) void ba_resident_kernel(RsArgs ra, int iters, int update_poses, int update_lms, int classify, int adaptive) {
    constexpr int kRsBlock = BLOCK, kRsWaves = BLOCK / 64, kRsMaxRows = 20 * kRsWaves;
    constexpr int kRsSub = kRsStreams / kRsWaves; // row streams a wave carries (1 or 2)
    const LmWindowArgs& a = ra.a;
    extern __shared__ __align__(16) unsigned char dyn[];
    __shared__ RsShared sm;
    const int w = ra.order ? ra.order[blockIdx.x] : (int)blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = a.n_kf_w ? min(max(a.n_kf_w[w], 1), a.n_kf) : a.n_kf, np = 6 * nk, nblk = nk * (nk + 1) / 2;
    const size_t Tbase = (size_t)w * a.n_kf * 7;
    const int lm0 = a.lm_off[w], nl = a.lm_off[w + 1] - lm0, e0 = a.edge_off[w], ne = a.edge_off[w + 1] - e0;
    const int nrows = (nl + 63) >> 6, nlp = nrows * 64;
    int n2 = 64;
    while (n2 < nl) n2 <<= 1;
    {   // does the window fit?  (uniform)
        const size_t need_run = (size_t)nblk * 288 + (size_t)nlp * 2, need_setup = (size_t)nlp * 6 + (size_t)n2 * 4; // (need_run: without the position rows, see rl0)
        // (dense graphs -- more than ~2.2 observations per landmark, e.g. the synthetic config-4 windows at 3.5 -- stay on lm_window_kernel: their
        // Schur work is hits, not landmarks, and its stored hit lists / weights win there: 3.78 vs 4.3 ms per 256 such windows)
        const bool dense = ra.dense_to_general && 5 * (long long)ne > 11 * (long long)nl;
        if (dense || nl <= 0 || ne <= 0 || nrows > 20 * (256 / 64) || n2 > kRsSortCap || need_run > (size_t)ra.dyn_narrow || need_setup > (size_t)ra.dyn_narrow) { // (the narrow form's limits, in both widths)
            if (tid == 0) ra.defer[w] = 1;
            return;
        }
        if (tid == 0) ra.defer[w] = 0;
    }
    long long* Sq = reinterpret_cast<long long*>(dyn);
    double* Sb = reinterpret_cast<double*>(dyn);
    // Positions: rows [rl0, nrows) in LDS (x[nlpL], y[nlpL], z[nlpL], nlpL = nlp - 64 rl0), rows [0, rl0) in the global working array Pw (same
    // role: the state the current phase works on; the accepted state during a trial is Pbak for both).  rl0 is decided after the setup (it needs the
    // first multi-observation landmark: only singles rows may live in global memory, the hit-major phase reads its landmarks from LDS).
    int rl0 = 0, sl0 = 0, nlpL = nlp;
    double* P = reinterpret_cast<double*>(dyn + (size_t)nblk * 288);
    unsigned short* live = nullptr; // (set with rl0)
    const float* xyz = a.xyz + 3 * (size_t)lm0;
    const int32_t* kfi = a.kf_idx + e0;
    const int32_t* lmi = a.lm_idx + e0;
    const float2* uv2 = reinterpret_cast<const float2*>(a.uv) + e0;
    float2* uvs = ra.uv_s + e0;
    int32_t* epos = ra.epos + e0;
    unsigned short* perm = reinterpret_cast<unsigned short*>(ra.tab + 6 * (size_t)lm0);
    unsigned short* mstat = perm + nl;
    float* xs = reinterpret_cast<float*>(ra.xin + 3 * (size_t)lm0);
    double* Pbak = ra.Pbak + 3 * (size_t)lm0;
    double* Dc = ra.Dc + 6 * (size_t)lm0;
    double* blc = ra.blc + 3 * (size_t)lm0;
    unsigned short* hit = nullptr; // sorted landmark per Schur hit, pair-major (set with rl0)
    int hit_cap = 0, hit_cap_narrow = 0;
    double* Pw = ra.tab + 6 * (size_t)lm0 + nl; // (the window's slice of `tab` is 6 nl doubles; perm / mstat take the first 4 nl bytes)
    const double K[4] = {a.K[0], a.K[1], a.K[2], a.K[3]};
    const CamK ck = make_camk(K);
    const double delta = a.huber_delta;
    const double f2sum = ck.fx2 + ck.fy2;
    const int slot27 = wave_slot<27>(lane), slot36 = wave_slot<36>(lane), slot33 = wave_slot<33>(lane), slot6 = wave_slot<6>(lane);
    int dst21 = 0, dst36 = 0; // where the butterfly sum this lane ends up with goes inside a 6 x 6 block
    if (slot33 >= 0 && slot33 < 21) { int rr = 0, rem = slot33; while (rem >= 6 - rr) { rem -= 6 - rr; ++rr; } dst21 = 6 * (rr + rem) + rr; } // upper (rr, cc) -> lower entry (cc, rr)
    if (slot36 >= 0) { const int rr = slot36 / 6, cc = slot36 - 6 * rr; dst36 = 6 * cc + rr; }
    long long* cyc = ra.dbg ? ra.dbg + kRsDbg * (size_t)w : nullptr;
    long long t_ph = cyc ? clock64() : 0;
#define RPH(i) do { if (cyc && tid == 0) { const long long t1__ = clock64(); atomicAdd(reinterpret_cast<unsigned long long*>(cyc) + (i), (unsigned long long)(t1__ - t_ph)); t_ph = t1__; } } while (0)

    // sum over the block of two per-lane values kept per row stream (v0[u], v1[u]: this wave's stream u = row stream wave + kRsWaves u):
    // deterministic and the same for both widths -- a butterfly per stream, the eight streams in order
    auto block_sum2 = [&](double (&v0)[kRsSub], double (&v1)[kRsSub], double& o0, double& o1) {
#pragma unroll
        for (int u = 0; u < kRsSub; ++u) { v0[u] = wave_sum(v0[u]); v1[u] = wave_sum(v1[u]); }
        __syncthreads();
        if (lane == 0) {
#pragma unroll
            for (int u = 0; u < kRsSub; ++u) { sm.red[2 * (wave + kRsWaves * u)] = v0[u]; sm.red[2 * (wave + kRsWaves * u) + 1] = v1[u]; }
        }
        __syncthreads();
        double s0 = 0, s1 = 0;
        for (int ww = 0; ww < kRsStreams; ++ww) { s0 += sm.red[2 * ww]; s1 += sm.red[2 * ww + 1]; }
        o0 = s0; o1 = s1;
    };
    auto stream_of = [&](int r) -> int { return (r / kRsWaves) % kRsSub; }; // which of this wave's streams row r (r mod kRsWaves == wave) belongs to (uniform)
    auto block_max1 = [&](double v) -> double {
        v = wave_max(v);
        __syncthreads();
        if (lane == 0) sm.red[wave] = v;
        __syncthreads();
        double s = sm.red[0];
        for (int ww = 1; ww < kRsWaves; ++ww) s = fmax(s, sm.red[ww]);
        return s;
    };

    // ------------------------------------------------------------------ setup (once per schedule): validate, sort, slot-major tables
    if (tid < 16) sm.flag[tid] = 0;
    if (tid < kRsKf + 2) sm.nq[tid] = 0;
    const int npairs = nk * (nk + 1) / 2;
    if (tid < npairs) {
        int k1 = 0, rem = tid;
        while (rem >= nk - k1) { rem -= nk - k1; ++k1; }
        sm.pk1[tid] = (unsigned char)k1; sm.pk2[tid] = (unsigned char)(k1 + rem);
    }
    {
        unsigned* t_mask = reinterpret_cast<unsigned*>(dyn);
        unsigned* keys = t_mask + nlp;
        unsigned short* inv = reinterpret_cast<unsigned short*>(keys + n2);
        for (int i = tid; i < nlp; i += kRsBlock) t_mask[i] = 0;
        __syncthreads();
        // keyframe set per landmark (bit k: observed by keyframe k), the keyframe of its LAST edge in bits 16..19
        for (int eb = tid; eb < ne; eb += 4 * kRsBlock) {
            int lv[4], lpv[4], lnv[4], kv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = min(eb + u * kRsBlock, ne - 1);
                lv[u] = lmi[e]; lpv[u] = e > 0 ? lmi[e - 1] : -1; lnv[u] = e + 1 < ne ? lmi[e + 1] : -1; kv[u] = kfi[e];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = eb + u * kRsBlock;
                if (e >= ne) break;
                const int l = lv[u], k = kv[u];
                if (l < lpv[u] || l < 0 || l >= nl || k < 0 || k >= nk) { sm.flag[7] = 1; continue; }
                const unsigned bit = 1u << k;
                const unsigned val = bit | (lnv[u] != l ? ((unsigned)k << 16) : 0u);
                const unsigned prev = atomicOr(&t_mask[l], val);
                if (prev & bit) sm.flag[7] = 2; // duplicate (keyframe, landmark) edge
            }
        }
        __syncthreads();
        if (sm.flag[7]) { // uniform
            if (tid == 0) { ra.status[w] = VSLAM_ERR_ARG; if (SCHED) ra.passes[w] = 0; }
            return;
        }
        // sort keys: count | keyframe set | landmark id: fewest observations first (the order vslam_build_windows_dev emits: its windows arrive sorted
        // and skip the sort below), equal sets adjacent, stable
        int sorted_ok = 1;
        for (int l = tid; l < n2; l += kRsBlock) {
            unsigned key = 0xFFFFFFFFu;
            if (l < nl) {
                const unsigned m = t_mask[l] & 0xFFFu;
                key = ((unsigned)__popc(m) << 25) | (m << 13) | (unsigned)l;
                if (l + 1 < nl) {
                    const unsigned m1 = t_mask[l + 1] & 0xFFFu;
                    const unsigned k1 = ((unsigned)__popc(m1) << 12) | m1;
                    if ((key >> 13) > k1) sorted_ok = 0;
                }
            }
            keys[l] = key;
        }
        const bool presorted = __syncthreads_and(sorted_ok) != 0;
        if (!presorted) {
            for (int k = 2; k <= n2; k <<= 1)
                for (int j = k >> 1; j > 0; j >>= 1) {
                    __syncthreads();
                    for (int t = tid; t < n2 / 2; t += kRsBlock) {
                        const int lo = 2 * t - (t & (j - 1)), hi = lo + j;
                        const bool up = (lo & k) == 0;
                        const unsigned x = keys[lo], y = keys[hi];
                        if ((x > y) == up) { keys[lo] = y; keys[hi] = x; }
                    }
                }
            __syncthreads();
        }
        // sorted tables; the landmarks with more than q observations are the SUFFIX [first_q, nl) of the sorted order (sm.nq[q] = nl - first_q of them)
        for (int s = tid; s < nl; s += kRsBlock) {
            const int l = (int)(keys[s] & 0x1FFFu);
            const unsigned m = t_mask[l];
            const unsigned m12 = m & 0xFFFu, lastk = (m >> 16) & 0xFu;
            const int c_here = __popc(m12);
            const int c_prev = s > 0 ? __popc(t_mask[keys[s - 1] & 0x1FFFu] & 0xFFFu) : 0;
            for (int q = c_prev; q < c_here; ++q) sm.nq[q] = nl - s;
            perm[s] = (unsigned short)l;
            mstat[s] = (unsigned short)(m12 | ((unsigned)__popc(m12 & ((1u << lastk) - 1u)) << 12));
            inv[l] = (unsigned short)s;
        }
        __syncthreads();
        if (tid == 0) {
            int acc = 0;
            // slot q of sorted landmark s sits at [slotoff[q] + s]: slot q's entries start where the earlier slots end, its first landmark is nl - n_q
            for (int q = 0; q <= kRsKf; ++q) { const int nqq = q < kRsKf ? sm.nq[q] : 0; sm.slotoff[q] = acc - (nl - nqq); acc += nqq; }
            if (acc != ne) sm.flag[7] = 3; // (cannot happen once duplicates are excluded; kept as a guard for the tables' bounds)
        }
        __syncthreads();
        if (sm.flag[7]) { if (tid == 0) { ra.status[w] = VSLAM_ERR_ARG; if (SCHED) ra.passes[w] = 0; } return; }
        for (int eb = tid; eb < ne; eb += 4 * kRsBlock) {
            int lv[4], kv[4]; float2 zv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int e = min(eb + u * kRsBlock, ne - 1); lv[u] = lmi[e]; kv[u] = kfi[e]; zv[u] = uv2[e]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = eb + u * kRsBlock;
                if (e >= ne) break;
                const unsigned m12 = t_mask[lv[u]] & 0xFFFu;
                const int q = __popc(m12 & ((1u << kv[u]) - 1u));
                const int pos = sm.slotoff[q] + inv[lv[u]];
                uvs[pos] = zv[u];
                if (ra.want_chi2) epos[pos] = e;
            }
        }
        // input positions in sorted order (every pass of the schedule starts from them)
        for (int s0 = tid; s0 < nl; s0 += 4 * kRsBlock) {
            int lq[4]; float v[4][3];
#pragma unroll
            for (int u = 0; u < 4; ++u) lq[u] = (int)(keys[min(s0 + u * kRsBlock, nl - 1)] & 0x1FFFu);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int c = 0; c < 3; ++c) v[u][c] = xyz[3 * lq[u] + c];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int s = s0 + u * kRsBlock;
                if (s < nl) { xs[3 * (size_t)s] = v[u][0]; xs[3 * (size_t)s + 1] = v[u][1]; xs[3 * (size_t)s + 2] = v[u][2]; }
            }
        }
        __syncthreads(); // (the tables are in global memory / sm; the dynamic region is free from here on)
    }
    {   // ---- which rows keep their positions in LDS (uniform).  Landmarks with several observations are the suffix [nl - nq[1], nl) of the sorted order.
        const int rm_static = (nl - sm.nq[1]) >> 6; // rows before it hold single-observation landmarks in every pass
        const long long fixed = (long long)nblk * 288 + (long long)nlp * 2;
        // the hit lists of the first pass (the longest: later passes only lose landmarks): a landmark with c >= 2 observations is named by c (c + 1) / 2 pairs
        long long hits = 0;
        for (int c = 2; c <= kRsKf; ++c) hits += (long long)(sm.nq[c - 1] - (c < kRsKf ? sm.nq[c] : 0)) * (c * (c + 1) / 2);
        const long long hit_bytes = min((2 * hits + 127) & ~63ll, (long long)kRsHitReserveMax);
        // decisions (take the window? hit lists?) against the NARROW form's LDS, the layout against this width's
        auto rows_for = [&](long long dynb, bool with_hits) -> int { return (int)max((dynb - fixed - (with_hits ? hit_bytes : 0)) / (64 * 24), 0ll); };
        int rows_n = rows_for(ra.dyn_narrow, true);
        if (nrows - rows_n > rm_static) rows_n = rows_for(ra.dyn_narrow, false); // (give up the hit lists -- the row-wise pair path -- before the window)
        const int rl0n = max(nrows - rows_n, 0);
        if (rl0n > rm_static) { // the multi-observation rows alone exceed the LDS: lm_window_kernel takes the window
            if (tid == 0) ra.defer[w] = 1;
            return;
        }
        hit_cap_narrow = (int)min(((long long)ra.dyn_narrow - (fixed + (long long)(nlp - 64 * rl0n) * 24)) / 2, (long long)0x7FFFFFF);
        int rows_lds = rows_for(ra.dyn_bytes, true);
        if (nrows - rows_lds > rm_static) rows_lds = rows_for(ra.dyn_bytes, false);
        rl0 = min(max(nrows - rows_lds, 0), rl0n);
        sl0 = 64 * rl0; nlpL = nlp - sl0;
        live = reinterpret_cast<unsigned short*>(dyn + (size_t)nblk * 288 + (size_t)nlpL * 24);
        hit = live + nlp;
        hit_cap = (int)min(((size_t)ra.dyn_bytes - ((size_t)nblk * 288 + (size_t)nlpL * 24 + (size_t)nlp * 2)) / 2, (size_t)0x7FFFFFF);
    }
    RPH(0);

    // ------------------------------------------------------------------ row helpers
    // A lane of row r owns sorted landmark s = 64 r + lane: live word (keyframe set | last slot << 12, 0 = not in the graph), position from
    // LDS, its first kRsSlotsReg observations from the slot-major table (one independent, coalesced load each).  The observation loads of
    // the NEXT row a wave will visit are issued before the current row is worked on (uv_issue / row_open): two waves per SIMD do not cover a
    // trip to L2 / HBM per row on their own (2.9 k cycles per row measured before the prefetch).
    // Rows [0, rm0) are SINGLES rows: every live landmark in them has exactly one observation (the sorted order puts them first).
    struct Row { int s; unsigned m; int lastq; double px, py, pz; float2 z[kRsSlotsReg]; unsigned U; int rc; };
    struct Pref { float2 z[kRsSlotsReg]; double p[3]; }; // what is requested a row ahead: the observations, and the position of a row that lives in global memory
    auto uv_issue = [&](int r, Pref& F) {
        if (r >= nrows) return;
        const int s = 64 * r + lane;
        const int rc = __builtin_amdgcn_readfirstlane((int)sm.rowC[r]);
        F.z[0] = uvs[max(min(sm.slotoff[0] + s, ne - 1), 0)];
#pragma unroll
        for (int q = 1; q < kRsSlotsReg; ++q)
            if (q < rc) F.z[q] = uvs[max(min(sm.slotoff[q] + s, ne - 1), 0)]; // (uniform branch)
        if (r < rl0) { const int sc = min(s, nl - 1); F.p[0] = Pw[sc]; F.p[1] = Pw[nl + sc]; F.p[2] = Pw[2 * (size_t)nl + sc]; } // (uniform branch)
    };
    auto row_open = [&](int r, const Pref& F, Row& R) {
        R.s = 64 * r + lane;
        const unsigned lv = live[R.s];
        R.m = lv & 0xFFFu; R.lastq = (int)(lv >> 12);
        R.U = __builtin_amdgcn_readfirstlane((unsigned)sm.rowU[r]);
        R.rc = __builtin_amdgcn_readfirstlane((int)sm.rowC[r]);
#pragma unroll
        for (int q = 0; q < kRsSlotsReg; ++q) R.z[q] = F.z[q];
        if (r < rl0) { R.px = F.p[0]; R.py = F.p[1]; R.pz = F.p[2]; }
        else { const int sl = R.s - sl0; R.px = P[sl]; R.py = P[nlpL + sl]; R.pz = P[2 * nlpL + sl]; }
    };
    auto store_pos = [&](int r, int sidx, double x, double y, double z) { // the working position of sorted landmark sidx (row r)
        if (r < rl0) { if (sidx < nl) { Pw[sidx] = x; Pw[nl + sidx] = y; Pw[2 * (size_t)nl + sidx] = z; } }
        else { const int sl = sidx - sl0; P[sl] = x; P[nlpL + sl] = y; P[2 * nlpL + sl] = z; }
    };
    auto obs_uv = [&](const Row& R, int q) -> float2 { // observation q of the lane's landmark (q < its count)
        float2 z = R.z[0];
#pragma unroll
        for (int i = 1; i < kRsSlotsReg; ++i) if (q == i) z = R.z[i];
        if (q >= kRsSlotsReg) z = uvs[max(min(sm.slotoff[min(q, kRsKf - 1)] + R.s, ne - 1), 0)];
        return z;
    };
    // |entry| bound of one observation's contributions to the normal equations (see the header): (fx^2 + fy^2) g^2 + w chi
    auto obs_bound = [&](double x, double y, double rho, double wchi) -> double {
        const double ax = fabs(x), ay = fabs(y), mx = fmax(ax, ay);
        const double g = fmax(fabs(rho) * fmax(1.0, mx), fma(mx, mx, 1.0));
        return fma(f2sum * g, g, wchi);
    };
    // A landmark with ONE observation couples nothing: its whole contribution to the reduced system is  At^T M At  with the 2 x 2 core
    //   M = L - L Bt (Bt^T L Bt + lambda I)^-1 Bt^T L = (L^-1 + Bt Bt^T / lambda)^-1 = lambda N,   N = (lambda L^-1 + G)^-1,
    //   G = Bt Bt^T = rho^2 [[1 + x^2, x y], [x y, 1 + y^2]]     (the rows of a rotation are orthonormal: no Jacobian, no 3 x 3 inverse)
    // and its step is  dx = -Bt^T N (en + At xp_k).  Most landmarks of a real sequence are like this (87 % on the bench's windows).
    auto single_core = [&](double x, double y, double rho, double wg, double lambda, double& n00, double& n01, double& n11) {
        const double iw = rcp_nr(wg), r2 = rho * rho;
        const double aa = fma(lambda * iw, ck.ifx * ck.ifx, r2 * fma(x, x, 1.0)), dd = fma(lambda * iw, ck.ify * ck.ify, r2 * fma(y, y, 1.0)), bb = r2 * x * y;
        const double idet = rcp_nr(aa * dd - bb * bb);
        n00 = dd * idet; n01 = -bb * idet; n11 = aa * idet;
    };

    // ---- evaluation at (Rsel, positions in LDS): robust cost and the fixed-point bound.  Static rows (row r -> wave r mod 8): fixed order.
    auto chi_pass = [&](const double* Rsel, double& bound_out) -> double {
        double part[kRsSub], bpart[kRsSub];
#pragma unroll
        for (int u = 0; u < kRsSub; ++u) { part[u] = 0; bpart[u] = 0; }
        Pref zn;
        uv_issue(wave, zn);
        for (int r = wave; r < nrows; r += kRsWaves) {
            Row R; row_open(r, zn, R);
            uv_issue(r + kRsWaves, zn);
            const int su = stream_of(r);
            double prow = 0, brow = 0;
            // (every lane walks ITS OWN observations, q-th with q-th: a row of mixed keyframe sets costs its longest track, not the size of the union)
            unsigned mm = R.m;
            for (int q = 0; q < R.rc; ++q) {
                if (mm) {
                    const int k = __builtin_ctz(mm); mm &= mm - 1;
                    const float2 z = obs_uv(R, q);
                    double x, y, rho, enx, eny, c, rob, wg;
                    cam_norm(&Rsel[12 * k], R.px, R.py, R.pz, x, y, rho);
                    eval_obs(ck, x, y, z, delta, enx, eny, c, rob, wg);
                    prow += rob;
                    brow += obs_bound(x, y, rho, wg * c);
                }
            }
#pragma unroll
            for (int u = 0; u < kRsSub; ++u) if (u == su) { part[u] += prow; bpart[u] += brow; }
        }
        double tot = 0;
        block_sum2(part, bpart, tot, bound_out);
        return tot;
    };

    // ---- one landmark-wise pass builds the whole reduced system at (sm.Rt, positions in LDS) for the given lambda.  Work items, drawn from an
    // LDS counter (costliest first): the rows from the last down to rm0 one by one (multi-observation landmarks: generic path below), then the singles
    // rows in chunks of kRsChunk.  diag_only: only diag(H_pp) (fixed point, sm.hdq) and the largest |H_ll| diagonal entry (returned per
    // thread) -- computeLambdaInit; every row takes the generic path then.
    auto linearise = [&](double lambda, double scale, bool diag_only) -> double {
        double maxdiag = 0;
        const bool hitmode = !diag_only && sm.flag[9] != 0;
        const int rm0 = diag_only ? 0 : sm.flag[8], nmulti = nrows - rm0;
        const int nitems = nmulti + (rm0 + kRsChunk - 1) / kRsChunk;
        auto item_row = [&](int it) -> int { return it < nmulti ? nrows - 1 - it : (it - nmulti) * kRsChunk; };
        auto draw = [&]() -> int {
            int it = 0;
            if (lane == 0) it = atomicAdd(&sm.flag[6], 1);
            return __builtin_amdgcn_readfirstlane(it);
        };
        Pref zn;
        int item = draw();
        if (item < nitems) uv_issue(item_row(item), zn);
        while (item < nitems) {
            const int next = draw();
            long long t_sub = cyc ? clock64() : 0;
#define RSUB(i) do { if (cyc && tid == 0) { const long long t1__ = clock64(); atomicAdd(reinterpret_cast<unsigned long long*>(cyc) + (i), (unsigned long long)(t1__ - t_sub)); t_sub = t1__; } } while (0)
            if (item >= nmulti) {
                // ---- singles chunk: contributions of consecutive rows accumulate in registers; folded when the keyframe changes
                const int r0 = item_row(item), r1 = min(r0 + kRsChunk, rm0);
                double acc[33];
#pragma unroll
                for (int i = 0; i < 33; ++i) acc[i] = 0;
                int kacc = -1;
                auto flush = [&](int k) {
                    wave_reduce_scatter<33>(acc, lane);
                    if (slot33 >= 0) {
                        if (!isfinite(acc[0])) sm.flag[1] = 1;
                        const unsigned long long qv = (unsigned long long)to_fixed(acc[0], scale);
                        if (slot33 < 21) atomicAdd(reinterpret_cast<unsigned long long*>(&Sq[rs_blk(k, k) + dst21]), qv);
                        else if (slot33 < 27) atomicAdd(reinterpret_cast<unsigned long long*>(&sm.bsq[6 * k + slot33 - 21]), qv);
                        else atomicAdd(reinterpret_cast<unsigned long long*>(&sm.bpq[6 * k + slot33 - 27]), qv);
                    }
#pragma unroll
                    for (int i = 0; i < 33; ++i) acc[i] = 0;
                };
                for (int r = r0; r < r1; ++r) {
                    Row R; row_open(r, zn, R);
                    uv_issue(r + 1 < r1 ? r + 1 : (next < nitems ? item_row(next) : nrows), zn);
                    unsigned U = R.U;
                    while (U) {
                        const int k = __builtin_ctz(U); U &= U - 1;
                        if (k != kacc) { if (kacc >= 0) flush(kacc); kacc = k; }
                        if (R.m == (1u << k)) {
                            double x, y, rho, enx, eny, c, rob, wg, A[12], n00, n01, n11;
                            cam_norm(&sm.Rt[12 * k], R.px, R.py, R.pz, x, y, rho);
                            eval_obs(ck, x, y, R.z[0], delta, enx, eny, c, rob, wg);
                            jac_norm(x, y, rho, A);
                            single_core(x, y, rho, wg, lambda, n00, n01, n11);
                            const double M00 = lambda * n00, M01 = lambda * n01, M11 = lambda * n11;
                            int idx = 0;
#pragma unroll
                            for (int rr = 0; rr < 6; ++rr) {
                                const double m0 = a_dot2(A, rr, M00, M01), m1 = a_dot2(A, rr, M01, M11);
#pragma unroll
                                for (int cc = rr; cc < 6; ++cc) { acc[idx] = a_fma2(A, cc, m0, m1, acc[idx]); ++idx; }
                            }
                            const double s0 = -(M00 * enx + M01 * eny), s1 = -(M01 * enx + M11 * eny);
                            const double p0 = -(wg * ck.fx2) * enx, p1 = -(wg * ck.fy2) * eny;
#pragma unroll
                            for (int rr = 0; rr < 6; ++rr) { acc[21 + rr] = a_fma2(A, rr, s0, s1, acc[21 + rr]); acc[27 + rr] = a_fma2(A, rr, p0, p1, acc[27 + rr]); }
                        }
                    }
                }
                if (kacc >= 0) flush(kacc);
                RSUB(14);
                item = next;
                continue;
            }
            // ---- generic row (landmarks with several observations)
            const int r = item_row(item);
            Row R; row_open(r, zn, R);
            if (next < nitems) uv_issue(item_row(next), zn);
            item = next;
            if (R.U == 0) continue;
            const bool on = R.m != 0;
            RSUB(10);
            // landmark blocks H_ll, b_l
            double h[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
            if (hitmode) { // (every lane walks its own observations; the pairs are formed hit-major below)
                unsigned mm = R.m;
                for (int q = 0; q < R.rc; ++q) {
                    if (mm) {
                        const int k = __builtin_ctz(mm); mm &= mm - 1;
                        const float2 z = obs_uv(R, q);
                        double x, y, rho, enx, eny, c, rob, wg, B[6], Bs[6];
                        const double* Rk = &sm.Rt[12 * k];
                        cam_norm(Rk, R.px, R.py, R.pz, x, y, rho);
                        eval_obs(ck, x, y, z, delta, enx, eny, c, rob, wg);
                        jac_point_norm(x, y, rho, Rk, B);
                        const double l0 = wg * ck.fx2, l1 = wg * ck.fy2;
#pragma unroll
                        for (int i = 0; i < 3; ++i) { Bs[i] = l0 * B[i]; Bs[3 + i] = l1 * B[3 + i]; }
                        h[0] += Bs[0] * B[0] + Bs[3] * B[3]; h[1] += Bs[0] * B[1] + Bs[3] * B[4]; h[2] += Bs[0] * B[2] + Bs[3] * B[5];
                        h[3] += Bs[1] * B[1] + Bs[4] * B[4]; h[4] += Bs[1] * B[2] + Bs[4] * B[5]; h[5] += Bs[2] * B[2] + Bs[5] * B[5];
                        g[0] -= Bs[0] * enx + Bs[3] * eny; g[1] -= Bs[1] * enx + Bs[4] * eny; g[2] -= Bs[2] * enx + Bs[5] * eny;
                    }
                }
            } else {
                unsigned U = R.U;
                while (U) {
                    const int k = __builtin_ctz(U); U &= U - 1;
                    const bool has = (R.m >> k) & 1u;
                    double hd[6] = {0, 0, 0, 0, 0, 0};
                    if (has) {
                        const int q = __popc(R.m & ((1u << k) - 1u));
                        const float2 z = obs_uv(R, q);
                        double x, y, rho, enx, eny, c, rob, wg, B[6], Bs[6];
                        const double* Rk = &sm.Rt[12 * k];
                        cam_norm(Rk, R.px, R.py, R.pz, x, y, rho);
                        eval_obs(ck, x, y, z, delta, enx, eny, c, rob, wg);
                        jac_point_norm(x, y, rho, Rk, B);
                        const double l0 = wg * ck.fx2, l1 = wg * ck.fy2;
#pragma unroll
                        for (int i = 0; i < 3; ++i) { Bs[i] = l0 * B[i]; Bs[3 + i] = l1 * B[3 + i]; }
                        h[0] += Bs[0] * B[0] + Bs[3] * B[3]; h[1] += Bs[0] * B[1] + Bs[3] * B[4]; h[2] += Bs[0] * B[2] + Bs[3] * B[5];
                        h[3] += Bs[1] * B[1] + Bs[4] * B[4]; h[4] += Bs[1] * B[2] + Bs[4] * B[5]; h[5] += Bs[2] * B[2] + Bs[5] * B[5];
                        g[0] -= Bs[0] * enx + Bs[3] * eny; g[1] -= Bs[1] * enx + Bs[4] * eny; g[2] -= Bs[2] * enx + Bs[5] * eny;
                        if (diag_only) {
                            double A[12];
                            jac_norm(x, y, rho, A);
#pragma unroll
                            for (int i = 0; i < 6; ++i) hd[i] = l0 * A[i] * A[i] + l1 * A[6 + i] * A[6 + i];
                        }
                    }
                    if (diag_only) { // diag(H_pp) of keyframe k: six sums over the row
                        wave_reduce_scatter<6>(hd, lane);
                        if (slot6 >= 0) {
                            if (!isfinite(hd[0])) sm.flag[1] = 1;
                            atomicAdd(reinterpret_cast<unsigned long long*>(&sm.hdq[6 * k + slot6]), (unsigned long long)to_fixed(hd[0], scale));
                        }
                    }
                }
            }
            if (on) maxdiag = fmax(maxdiag, fmax(fabs(h[0]), fmax(fabs(h[3]), fabs(h[5]))));
            RSUB(11);
            if (diag_only) continue;
            double Di[6] = {0, 0, 0, 0, 0, 0};
            if (on && !inv3_sym(h[0] + lambda, h[1], h[2], h[3] + lambda, h[4], h[5] + lambda, Di)) sm.flag[1] = 1;
            if (hitmode) { // the pairs are formed hit-major (below): leave D^-1 and b_l of the row's landmarks behind
                if (on) {
                    double2* dq = reinterpret_cast<double2*>(Dc + 6 * (size_t)R.s);
                    dq[0] = make_double2(Di[0], Di[1]); dq[1] = make_double2(Di[2], Di[3]); dq[2] = make_double2(Di[4], Di[5]);
                    blc[3 * (size_t)R.s] = g[0]; blc[3 * (size_t)R.s + 1] = g[1]; blc[3 * (size_t)R.s + 2] = g[2];
                }
                continue;
            }
            // pairs of keyframes of the row
            unsigned U1 = R.U;
            while (U1) {
                const int k1 = __builtin_ctz(U1); U1 &= U1 - 1;
                const bool has1 = (R.m >> k1) & 1u;
                double A1[12], BD[6], en1x = 0, en1y = 0, l10 = 0, l11 = 0;
#pragma unroll
                for (int i = 0; i < 12; ++i) A1[i] = 0;
#pragma unroll
                for (int i = 0; i < 6; ++i) BD[i] = 0;
                double B1[6] = {0, 0, 0, 0, 0, 0};
                if (has1) {
                    const int q = __popc(R.m & ((1u << k1) - 1u));
                    const float2 z = obs_uv(R, q);
                    double x, y, rho, c, rob, wg;
                    const double* Rk = &sm.Rt[12 * k1];
                    cam_norm(Rk, R.px, R.py, R.pz, x, y, rho);
                    eval_obs(ck, x, y, z, delta, en1x, en1y, c, rob, wg);
                    jac_norm(x, y, rho, A1);
                    jac_point_norm(x, y, rho, Rk, B1);
                    l10 = wg * ck.fx2; l11 = wg * ck.fy2;
#pragma unroll
                    for (int r2 = 0; r2 < 2; ++r2) { // BD = Bt D^-1 (2 x 3)
                        BD[3 * r2] = B1[3 * r2] * Di[0] + B1[3 * r2 + 1] * Di[1] + B1[3 * r2 + 2] * Di[2];
                        BD[3 * r2 + 1] = B1[3 * r2] * Di[1] + B1[3 * r2 + 1] * Di[3] + B1[3 * r2 + 2] * Di[4];
                        BD[3 * r2 + 2] = B1[3 * r2] * Di[2] + B1[3 * r2 + 1] * Di[4] + B1[3 * r2 + 2] * Di[5];
                    }
                }
                {   // diagonal block: At^T (L - L N L) At, N = Bt D^-1 Bt^T; right-hand sides: b_p = -At^T L en, b_s = b_p - At^T L (BD b_l)
                    const double N00 = BD[0] * B1[0] + BD[1] * B1[1] + BD[2] * B1[2];
                    const double N01 = BD[0] * B1[3] + BD[1] * B1[4] + BD[2] * B1[5];
                    const double N11 = BD[3] * B1[3] + BD[4] * B1[4] + BD[5] * B1[5];
                    const double M00 = l10 - l10 * l10 * N00, M01 = -l10 * l11 * N01, M11 = l11 - l11 * l11 * N11;
                    double red[33];
                    int idx = 0;
#pragma unroll
                    for (int rr = 0; rr < 6; ++rr) {
                        const double m0 = a_dot2(A1, rr, M00, M01), m1 = a_dot2(A1, rr, M01, M11);
#pragma unroll
                        for (int cc = rr; cc < 6; ++cc) red[idx++] = a_fma2(A1, cc, m0, m1, 0.0);
                    }
                    const double v0 = BD[0] * g[0] + BD[1] * g[1] + BD[2] * g[2], v1 = BD[3] * g[0] + BD[4] * g[1] + BD[5] * g[2];
                    const double p0 = -l10 * en1x, p1 = -l11 * en1y;
                    const double s0 = p0 - l10 * v0, s1 = p1 - l11 * v1;
#pragma unroll
                    for (int rr = 0; rr < 6; ++rr) { red[21 + rr] = a_fma2(A1, rr, s0, s1, 0.0); red[27 + rr] = a_fma2(A1, rr, p0, p1, 0.0); }
                    wave_reduce_scatter<33>(red, lane);
                    if (slot33 >= 0) {
                        if (!isfinite(red[0])) sm.flag[1] = 1;
                        const unsigned long long qv = (unsigned long long)to_fixed(red[0], scale);
                        if (slot33 < 21) atomicAdd(reinterpret_cast<unsigned long long*>(&Sq[rs_blk(k1, k1) + dst21]), qv);
                        else if (slot33 < 27) atomicAdd(reinterpret_cast<unsigned long long*>(&sm.bsq[6 * k1 + slot33 - 21]), qv);
                        else atomicAdd(reinterpret_cast<unsigned long long*>(&sm.bpq[6 * k1 + slot33 - 27]), qv);
                    }
                }
                RSUB(12);
                unsigned U2 = U1; // keyframes after k1
                while (U2) {
                    const int k2 = __builtin_ctz(U2); U2 &= U2 - 1;
                    const bool both = has1 && ((R.m >> k2) & 1u);
                    if (__ballot(both) == 0ull) continue; // no landmark of the row sees both
                    double acc[36];
#pragma unroll
                    for (int i = 0; i < 36; ++i) acc[i] = 0;
                    if (both) {
                        const int q = __popc(R.m & ((1u << k2) - 1u));
                        const float2 z = obs_uv(R, q);
                        double x, y, rho, enx, eny, c, rob, wg, A2[12], B2[6];
                        const double* Rk = &sm.Rt[12 * k2];
                        cam_norm(Rk, R.px, R.py, R.pz, x, y, rho);
                        eval_obs(ck, x, y, z, delta, enx, eny, c, rob, wg);
                        jac_norm(x, y, rho, A2);
                        jac_point_norm(x, y, rho, Rk, B2);
                        const double l20 = wg * ck.fx2, l21 = wg * ck.fy2;
                        // M = L1 (BD Bt2^T) L2 (2 x 2); block (k1, k2) = -At1^T M At2
                        double M[4];
                        M[0] = -l10 * l20 * (BD[0] * B2[0] + BD[1] * B2[1] + BD[2] * B2[2]);
                        M[1] = -l10 * l21 * (BD[0] * B2[3] + BD[1] * B2[4] + BD[2] * B2[5]);
                        M[2] = -l11 * l20 * (BD[3] * B2[0] + BD[4] * B2[1] + BD[5] * B2[2]);
                        M[3] = -l11 * l21 * (BD[3] * B2[3] + BD[4] * B2[4] + BD[5] * B2[5]);
#pragma unroll
                        for (int rr = 0; rr < 6; ++rr) {
                            const double m0 = a_dot2(A1, rr, M[0], M[2]), m1 = a_dot2(A1, rr, M[1], M[3]);
#pragma unroll
                            for (int cc = 0; cc < 6; ++cc) acc[6 * rr + cc] = a_fma2(A2, cc, m0, m1, 0.0);
                        }
                    }
                    wave_reduce_scatter<36>(acc, lane);
                    if (slot36 >= 0) {
                        if (!isfinite(acc[0])) sm.flag[1] = 1;
                        // S[6 k1 + rr][6 k2 + cc] -> lower entry (k2, k1)(cc, rr)
                        atomicAdd(reinterpret_cast<unsigned long long*>(&Sq[rs_blk(k2, k1) + dst36]), (unsigned long long)to_fixed(acc[0], scale));
                    }
                }
                RSUB(13);
            }
        }
        if (hitmode) {
            // ---- Schur blocks of the multi-observation landmarks, HIT-MAJOR: the list of keyframe pair (k1, k2) names every landmark that
            // sees both (packed: all 64 lanes carry a hit, where a row of 64 landmarks of mixed keyframe sets kept a handful busy per pair);
            // a work item = up to kRsHitChunk rows of one pair's list, summed in registers, one butterfly, integer atomics.
            __syncthreads(); // (D^-1, b_l of every multi-observation landmark are in place)
            const int nhit_items = sm.itemoff[npairs];
            for (;;) {
                int item = 0;
                if (lane == 0) item = atomicAdd(&sm.flag[10], 1);
                item = __builtin_amdgcn_readfirstlane(item);
                if (item >= nhit_items) break;
                const int p = sm.itempair[item];
                const int k1 = sm.pk1[p], k2 = sm.pk2[p];
                const int jbeg = sm.pairoff[p] + (item - sm.itemoff[p]) * (kRsHitChunk * 64), jend = min(sm.pairoff[p + 1], jbeg + kRsHitChunk * 64);
                const double* R1 = &sm.Rt[12 * k1];
                const double* R2 = &sm.Rt[12 * k2];
                double acc[36];
#pragma unroll
                for (int i = 0; i < 36; ++i) acc[i] = 0;
                for (int j = jbeg; j < jend; j += 64) {
                    const bool valid = j + lane < jend;
                    const int s = hit[min(j + lane, jend - 1)];
                    const unsigned m = live[s] & 0xFFFu;
                    const int q1 = __popc(m & ((1u << k1) - 1u)), q2 = __popc(m & ((1u << k2) - 1u));
                    const float2 z1 = uvs[max(min(sm.slotoff[min(q1, kRsKf - 1)] + s, ne - 1), 0)];
                    const float2 z2 = uvs[max(min(sm.slotoff[min(q2, kRsKf - 1)] + s, ne - 1), 0)];
                    const double2* dq = reinterpret_cast<const double2*>(Dc + 6 * (size_t)s);
                    const double2 Da = dq[0], Db = dq[1], Dcc = dq[2];
                    const double px = P[s - sl0], py = P[nlpL + s - sl0], pz = P[2 * nlpL + s - sl0]; // (a hit's landmark has several observations: its row is in LDS)
                    double g0 = 0, g1 = 0, g2 = 0;
                    if (k1 == k2) { g0 = blc[3 * (size_t)s]; g1 = blc[3 * (size_t)s + 1]; g2 = blc[3 * (size_t)s + 2]; }
                    if (!valid) continue;
                    double x, y, rho, en1x, en1y, c, rob, wg, A1[12], B1[6], BD[6];
                    cam_norm(R1, px, py, pz, x, y, rho);
                    eval_obs(ck, x, y, z1, delta, en1x, en1y, c, rob, wg);
                    jac_norm(x, y, rho, A1);
                    jac_point_norm(x, y, rho, R1, B1);
                    const double l10 = wg * ck.fx2, l11 = wg * ck.fy2;
#pragma unroll
                    for (int r2 = 0; r2 < 2; ++r2) {
                        BD[3 * r2] = B1[3 * r2] * Da.x + B1[3 * r2 + 1] * Da.y + B1[3 * r2 + 2] * Db.x;
                        BD[3 * r2 + 1] = B1[3 * r2] * Da.y + B1[3 * r2 + 1] * Db.y + B1[3 * r2 + 2] * Dcc.x;
                        BD[3 * r2 + 2] = B1[3 * r2] * Db.x + B1[3 * r2 + 1] * Dcc.x + B1[3 * r2 + 2] * Dcc.y;
                    }
                    if (k1 == k2) { // (uniform) diagonal pair: At^T (L - L N L) At and the right-hand sides
                        const double N00 = BD[0] * B1[0] + BD[1] * B1[1] + BD[2] * B1[2];
                        const double N01 = BD[0] * B1[3] + BD[1] * B1[4] + BD[2] * B1[5];
                        const double N11 = BD[3] * B1[3] + BD[4] * B1[4] + BD[5] * B1[5];
                        const double M00 = l10 - l10 * l10 * N00, M01 = -l10 * l11 * N01, M11 = l11 - l11 * l11 * N11;
                        int idx = 0;
#pragma unroll
                        for (int rr = 0; rr < 6; ++rr) {
                            const double m0 = a_dot2(A1, rr, M00, M01), m1 = a_dot2(A1, rr, M01, M11);
#pragma unroll
                            for (int cc = rr; cc < 6; ++cc) { acc[idx] = a_fma2(A1, cc, m0, m1, acc[idx]); ++idx; }
                        }
                        const double v0 = BD[0] * g0 + BD[1] * g1 + BD[2] * g2, v1 = BD[3] * g0 + BD[4] * g1 + BD[5] * g2;
                        const double p0 = -l10 * en1x, p1 = -l11 * en1y;
                        const double s0 = p0 - l10 * v0, s1 = p1 - l11 * v1;
#pragma unroll
                        for (int rr = 0; rr < 6; ++rr) { acc[21 + rr] = a_fma2(A1, rr, s0, s1, acc[21 + rr]); acc[27 + rr] = a_fma2(A1, rr, p0, p1, acc[27 + rr]); }
                    } else {
                        double enx, eny, A2[12], B2[6];
                        cam_norm(R2, px, py, pz, x, y, rho);
                        eval_obs(ck, x, y, z2, delta, enx, eny, c, rob, wg);
                        jac_norm(x, y, rho, A2);
                        jac_point_norm(x, y, rho, R2, B2);
                        const double l20 = wg * ck.fx2, l21 = wg * ck.fy2;
                        double M[4];
                        M[0] = -l10 * l20 * (BD[0] * B2[0] + BD[1] * B2[1] + BD[2] * B2[2]);
                        M[1] = -l10 * l21 * (BD[0] * B2[3] + BD[1] * B2[4] + BD[2] * B2[5]);
                        M[2] = -l11 * l20 * (BD[3] * B2[0] + BD[4] * B2[1] + BD[5] * B2[2]);
                        M[3] = -l11 * l21 * (BD[3] * B2[3] + BD[4] * B2[4] + BD[5] * B2[5]);
#pragma unroll
                        for (int rr = 0; rr < 6; ++rr) {
                            const double m0 = a_dot2(A1, rr, M[0], M[2]), m1 = a_dot2(A1, rr, M[1], M[3]);
#pragma unroll
                            for (int cc = 0; cc < 6; ++cc) acc[6 * rr + cc] = a_fma2(A2, cc, m0, m1, acc[6 * rr + cc]);
                        }
                    }
                }
                if (k1 == k2) {
                    double red[33];
#pragma unroll
                    for (int i = 0; i < 33; ++i) red[i] = acc[i];
                    wave_reduce_scatter<33>(red, lane);
                    if (slot33 >= 0) {
                        if (!isfinite(red[0])) sm.flag[1] = 1;
                        const unsigned long long qv = (unsigned long long)to_fixed(red[0], scale);
                        if (slot33 < 21) atomicAdd(reinterpret_cast<unsigned long long*>(&Sq[rs_blk(k1, k1) + dst21]), qv);
                        else if (slot33 < 27) atomicAdd(reinterpret_cast<unsigned long long*>(&sm.bsq[6 * k1 + slot33 - 21]), qv);
                        else atomicAdd(reinterpret_cast<unsigned long long*>(&sm.bpq[6 * k1 + slot33 - 27]), qv);
                    }
                } else {
                    wave_reduce_scatter<36>(acc, lane);
                    if (slot36 >= 0) {
                        if (!isfinite(acc[0])) sm.flag[1] = 1;
                        atomicAdd(reinterpret_cast<unsigned long long*>(&Sq[rs_blk(k2, k1) + dst36]), (unsigned long long)to_fixed(acc[0], scale));
                    }
                }
            }
        }
#undef RSUB
        return maxdiag;
    };

    // ---- back-substitution + trial evaluation in one visit: D^-1, b_l re-derived at the accepted state (sm.Rt, LDS positions), the
    // position moved IN PLACE, the robust cost and the bound at the trial state (sm.RtT).  Static rows.
    auto backsub_pass = [&](double lambda, bool backup, double& scale_out, double& bound_out) -> double {
        double part[kRsSub], bpart[kRsSub], spart[kRsSub], zero_[kRsSub];
#pragma unroll
        for (int u = 0; u < kRsSub; ++u) { part[u] = 0; bpart[u] = 0; spart[u] = 0; zero_[u] = 0; }
        const int rm0 = sm.flag[8];
        Pref zn;
        uv_issue(wave, zn);
        for (int r = wave; r < nrows; r += kRsWaves) {
            Row R; row_open(r, zn, R);
            uv_issue(r + kRsWaves, zn);
            if (R.U == 0) continue;
            const int su = stream_of(r);
            double prow = 0, brow = 0, srow = 0;
            auto fold_row = [&]() {
#pragma unroll
                for (int u = 0; u < kRsSub; ++u) if (u == su) { part[u] += prow; bpart[u] += brow; spart[u] += srow; }
            };
            const bool on = R.m != 0;
            if (backup && R.s < nl) { Pbak[R.s] = R.px; Pbak[nl + R.s] = R.py; Pbak[2 * (size_t)nl + R.s] = R.pz; }
            if (r < rm0) { // singles row
                if (on) {
                    const int k = __builtin_ctz(R.m);
                    double x, y, rho, enx, eny, c, rob, wg, A[12], B[6], n00, n01, n11;
                    const double* Rk = &sm.Rt[12 * k];
                    cam_norm(Rk, R.px, R.py, R.pz, x, y, rho);
                    eval_obs(ck, x, y, R.z[0], delta, enx, eny, c, rob, wg);
                    jac_norm(x, y, rho, A);
                    jac_point_norm(x, y, rho, Rk, B);
                    single_core(x, y, rho, wg, lambda, n00, n01, n11);
                    double v0 = enx, v1 = eny; // en + At xp_k
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        const double xr = sm.xp[6 * k + i];
                        if (i != 1) v0 = fma(A[i], xr, v0);
                        if (i != 0) v1 = fma(A[6 + i], xr, v1);
                    }
                    const double m0 = n00 * v0 + n01 * v1, m1 = n01 * v0 + n11 * v1;
                    const double l0e = wg * ck.fx2 * enx, l1e = wg * ck.fy2 * eny;
                    double dx[3];
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        dx[i] = -(B[i] * m0 + B[3 + i] * m1);
                        const double bl = -(B[i] * l0e + B[3 + i] * l1e);
                        srow += dx[i] * (lambda * dx[i] + bl);
                    }
                    R.px += dx[0]; R.py += dx[1]; R.pz += dx[2];
                    store_pos(r, R.s, R.px, R.py, R.pz);
                    cam_norm(&sm.RtT[12 * k], R.px, R.py, R.pz, x, y, rho);
                    eval_obs(ck, x, y, R.z[0], delta, enx, eny, c, rob, wg);
                    prow += rob;
                    brow += obs_bound(x, y, rho, wg * c);
                }
                fold_row();
                continue;
            }
            double h[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0}, cacc[3] = {0, 0, 0};
            {
                unsigned mm = R.m;
                for (int q = 0; q < R.rc; ++q) {
                    if (mm) {
                        const int k = __builtin_ctz(mm); mm &= mm - 1;
                        const float2 z = obs_uv(R, q);
                        double x, y, rho, enx, eny, c, rob, wg, A[12], B[6], Bs[6];
                        const double* Rk = &sm.Rt[12 * k];
                        cam_norm(Rk, R.px, R.py, R.pz, x, y, rho);
                        eval_obs(ck, x, y, z, delta, enx, eny, c, rob, wg);
                        jac_norm(x, y, rho, A);
                        jac_point_norm(x, y, rho, Rk, B);
                        const double l0 = wg * ck.fx2, l1 = wg * ck.fy2;
#pragma unroll
                        for (int i = 0; i < 3; ++i) { Bs[i] = l0 * B[i]; Bs[3 + i] = l1 * B[3 + i]; }
                        h[0] += Bs[0] * B[0] + Bs[3] * B[3]; h[1] += Bs[0] * B[1] + Bs[3] * B[4]; h[2] += Bs[0] * B[2] + Bs[3] * B[5];
                        h[3] += Bs[1] * B[1] + Bs[4] * B[4]; h[4] += Bs[1] * B[2] + Bs[4] * B[5]; h[5] += Bs[2] * B[2] + Bs[5] * B[5];
                        g[0] -= Bs[0] * enx + Bs[3] * eny; g[1] -= Bs[1] * enx + Bs[4] * eny; g[2] -= Bs[2] * enx + Bs[5] * eny;
                        double a0 = 0, a1 = 0; // At xp_k
#pragma unroll
                        for (int i = 0; i < 6; ++i) {
                            const double xr = sm.xp[6 * k + i];
                            if (i != 1) a0 = fma(A[i], xr, a0);
                            if (i != 0) a1 = fma(A[6 + i], xr, a1);
                        }
                        cacc[0] -= Bs[0] * a0 + Bs[3] * a1; cacc[1] -= Bs[1] * a0 + Bs[4] * a1; cacc[2] -= Bs[2] * a0 + Bs[5] * a1;
                    }
                }
            }
            if (on) {
                double Di[6];
                inv3_sym(h[0] + lambda, h[1], h[2], h[3] + lambda, h[4], h[5] + lambda, Di);
                const double c0 = g[0] + cacc[0], c1 = g[1] + cacc[1], c2 = g[2] + cacc[2];
                const double x0 = Di[0] * c0 + Di[1] * c1 + Di[2] * c2;
                const double x1 = Di[1] * c0 + Di[3] * c1 + Di[4] * c2;
                const double x2 = Di[2] * c0 + Di[4] * c1 + Di[5] * c2;
                srow += x0 * (lambda * x0 + g[0]) + x1 * (lambda * x1 + g[1]) + x2 * (lambda * x2 + g[2]);
                R.px += x0; R.py += x1; R.pz += x2;
                store_pos(r, R.s, R.px, R.py, R.pz);
            }
            {
                unsigned mm = R.m;
                for (int q = 0; q < R.rc; ++q) {
                    if (mm) {
                        const int k = __builtin_ctz(mm); mm &= mm - 1;
                        const float2 z = obs_uv(R, q);
                        double x, y, rho, enx, eny, c, rob, wg;
                        cam_norm(&sm.RtT[12 * k], R.px, R.py, R.pz, x, y, rho);
                        eval_obs(ck, x, y, z, delta, enx, eny, c, rob, wg);
                        prow += rob;
                        brow += obs_bound(x, y, rho, wg * c);
                    }
                }
            }
            fold_row();
        }
        if (tid < np) spart[0] += sm.xp[tid] * (lambda * sm.xp[tid] + sm.bp[tid]); // (lanes of wave 0: row stream 0 in both widths)
        double tot = 0, stot = 0, dummy = 0;
        block_sum2(part, bpart, tot, bound_out);
        block_sum2(spart, zero_, stot, dummy);
        scale_out = stot + 1e-3;
        return tot;
    };
    auto restore_positions = [&]() {
        for (int s = tid; s < nl; s += kRsBlock) store_pos(s >> 6, s, Pbak[s], Pbak[nl + s], Pbak[2 * (size_t)nl + s]);
        __syncthreads();
    };

    // ---- pass init.  mode 0: first optimize_map pass (live = inlier && reliable_depth_, optimization.cpp:160); 1: a later pass (what the classification
    // left).  Positions from the sorted input copy, row unions, singles rows, hit lists, poses.
    auto pass_init = [&](int mode) {
        __syncthreads();
        for (int s0 = tid; s0 < nlp; s0 += 4 * kRsBlock) {
            unsigned lvv[4]; float v[4][3];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int s = s0 + u * kRsBlock;
                lvv[u] = 0;
#pragma unroll
                for (int c = 0; c < 3; ++c) v[u][c] = 0.f;
                if (s < nl) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) v[u][c] = xs[3 * (size_t)s + c];
                    if (mode != 1) {
                        const unsigned ms = mstat[s];
                        const int l = perm[s];
                        const uint8_t inl = a.lm_inlier[lm0 + l];
                        const uint8_t rel = a.reliable ? a.reliable[lm0 + l] : (uint8_t)1;
                        lvv[u] = ((ms & 0xFFFu) != 0 && inl != 0 && rel != 0) ? ms : 0u;
                    } else lvv[u] = live[s];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int s = s0 + u * kRsBlock;
                if (s >= nlp) break; // (uniform per wave: s0 is 64-aligned per wave)
                live[s] = (unsigned short)lvv[u];
                store_pos(s >> 6, s, (double)v[u][0], (double)v[u][1], (double)v[u][2]);
                unsigned um = lvv[u] & 0xFFFu;
                int cm = __popc(um);
                for (int o = 32; o > 0; o >>= 1) { um |= __shfl_xor(um, o); cm = max(cm, __shfl_xor(cm, o)); }
                if (lane == 0) { sm.rowU[s >> 6] = (unsigned short)um; sm.rowC[s >> 6] = (unsigned char)cm; }
            }
        }
        for (int i = tid; i < nk * 7; i += kRsBlock) sm.T[i] = a.T[Tbase + i];
        __syncthreads();
        if (tid == kRsBlock - 1) { // singles rows: the leading run of rows whose live landmarks all have exactly one observation
            int r = 0;
            while (r < nrows && sm.rowC[r] <= 1) ++r;
            sm.flag[8] = r;
        }
        if (tid < nk) expand_pose(&sm.T[7 * tid], &sm.Rt[12 * tid]);
        __syncthreads();
        {   // ---- hit lists of the multi-observation rows [rm0, nrows): per keyframe pair (k1 <= k2) the sorted landmarks that are live and see both,
            // in sorted order (count, prefix, write: a wave takes the pairs p = wave, wave + 8, ...).  Too many hits for the LDS the window leaves
            // free (or no such rows): flag[9] = 0, the rows then form their pairs themselves (row-wise path of the linearisation).
            const int rm0 = sm.flag[8];
            for (int pass2 = 0; pass2 < 2; ++pass2) {
                if (pass2 == 1 && !sm.flag[9]) break; // (uniform)
                for (int p = wave; p < npairs; p += kRsWaves) {
                    const int k1 = sm.pk1[p], k2 = sm.pk2[p];
                    int run = pass2 ? sm.pairoff[p] : 0;
                    for (int r = rm0; r < nrows; ++r) {
                        const unsigned U = sm.rowU[r];
                        if (!((U >> k1) & 1u) || !((U >> k2) & 1u)) continue; // (uniform)
                        const unsigned m = live[64 * r + lane] & 0xFFFu;
                        const bool both = ((m >> k1) & 1u) && ((m >> k2) & 1u);
                        const unsigned long long b = __ballot(both);
                        if (pass2 && both) hit[run + __popcll(b & ((1ull << lane) - 1ull))] = (unsigned short)(64 * r + lane);
                        run += __popcll(b);
                    }
                    if (!pass2 && lane == 0) sm.pairoff[p + 1] = run; // (count, turned into offsets below)
                }
                __syncthreads();
                if (!pass2) {
                    if (tid == 0) {
                        int acc = 0, items = 0;
                        sm.itemoff[0] = 0;
                        for (int p = 0; p < npairs; ++p) {
                            const int c = sm.pairoff[p + 1];
                            sm.pairoff[p] = acc; acc += c;
                            const int ni = (c + kRsHitChunk * 64 - 1) / (kRsHitChunk * 64);
                            for (int i = 0; i < ni && items + i < kRsHitItems; ++i) sm.itempair[items + i] = (unsigned char)p;
                            items += ni;
                            sm.itemoff[p + 1] = items;
                        }
                        sm.pairoff[npairs] = acc;
                        sm.flag[9] = (rm0 < nrows && acc > 0 && acc <= min(hit_cap, hit_cap_narrow) && items <= kRsHitItems) ? 1 : 0; // (the narrow form's capacity decides in both widths)
                    }
                    __syncthreads();
                }
            }
        }
    };

    // ---- chi2 of every active edge at (Rsel, LDS positions), the adaptive threshold and the landmark flags (optimization.cpp:224-266, :395-424): the
    // landmark's last edge decides.  Returns whether this thread cleared a flag.
    auto classify_pass = [&](const double* Rsel) -> int {
        int newly_flagged = 0;
            // per row slot of this wave (row = wave + 8 j): six bits "chi2 of the landmark's last edge > delta 2^i", i = 0..5
            unsigned long long lastbits_lo = 0, lastbits_hi = 0;
            int cnt_out[5] = {0, 0, 0, 0, 0}, cnt_all = 0;
            double thv[6];
            thv[0] = delta;
#pragma unroll
            for (int i = 1; i < 6; ++i) thv[i] = thv[i - 1] * 2;
            double* chi2 = (ra.want_chi2 && a.chi2) ? a.chi2 + e0 : nullptr;
            if (chi2) { for (int e = tid; e < ne; e += kRsBlock) chi2[e] = 0.0; __syncthreads(); }
            Pref zn;
            uv_issue(wave, zn);
            for (int j = 0; wave + kRsWaves * j < nrows; ++j) {
                const int r = wave + kRsWaves * j;
                Row R; row_open(r, zn, R);
                uv_issue(r + kRsWaves, zn);
                unsigned mm = R.m, bits = 0;
                for (int q = 0; q < R.rc; ++q) {
                    if (mm) {
                        const int k = __builtin_ctz(mm); mm &= mm - 1;
                        const float2 z = obs_uv(R, q);
                        double X, Y, Z, ex, ey;
                        project_err(&Rsel[12 * k], K, R.px, R.py, R.pz, z.x, z.y, X, Y, Z, ex, ey);
                        const double c = ex * ex + ey * ey;
                        ++cnt_all;
#pragma unroll
                        for (int i = 0; i < 5; ++i) cnt_out[i] += c > thv[i];
                        if (q == R.lastq) {
#pragma unroll
                            for (int i = 0; i < 6; ++i) bits |= (c > thv[i]) ? (1u << i) : 0u;
                        }
                        if (chi2) chi2[epos[sm.slotoff[min(q, kRsKf - 1)] + R.s]] = c;
                    }
                }
                if (j < 10) lastbits_lo |= (unsigned long long)bits << (6 * j); else lastbits_hi |= (unsigned long long)bits << (6 * (j - 10));
            }
            if (cyc && tid == 0) { const long long t1__ = clock64(); atomicAdd(reinterpret_cast<unsigned long long*>(cyc) + 15, (unsigned long long)(t1__ - t_ph)); }
            double th = delta;
            if (classify) {
                int v[6] = {cnt_out[0], cnt_out[1], cnt_out[2], cnt_out[3], cnt_out[4], cnt_all};
#pragma unroll
                for (int i = 0; i < 6; ++i) for (int o = 32; o > 0; o >>= 1) v[i] += __shfl_xor(v[i], o);
                __syncthreads();
                if (lane == 0)
#pragma unroll
                    for (int i = 0; i < 6; ++i) sm.redi[8 * wave + i] = v[i];
                __syncthreads();
                int tot[6] = {0, 0, 0, 0, 0, 0};
                for (int ww = 0; ww < kRsWaves; ++ww)
#pragma unroll
                    for (int i = 0; i < 6; ++i) tot[i] += sm.redi[8 * ww + i];
                int ti = 0;
                for (int iteration = 0; iteration < 5; ++iteration) {
                    const double out = (double)tot[iteration], in = (double)(tot[5] - tot[iteration]);
                    const double ratio = in / (in + out);
                    if (ratio > 0.5) break;
                    th *= 2; ++ti;
                }
                for (int j = 0; wave + kRsWaves * j < nrows; ++j) {
                    const int s = 64 * (wave + kRsWaves * j) + lane;
                    const unsigned lv = live[s];
                    if (lv & 0xFFFu) {
                        const unsigned b6 = j < 10 ? (unsigned)(lastbits_lo >> (6 * j)) : (unsigned)(lastbits_hi >> (6 * (j - 10)));
                        const bool keep = ((b6 >> ti) & 1u) == 0;
                        if (!keep) { a.lm_inlier[lm0 + perm[s]] = 0; live[s] = 0; newly_flagged = 1; } // (a live landmark's flag is 1: only a change is written)
                    }
                }
                if (tid == 0 && a.chi2_thr) a.chi2_thr[w] = th;
            }
        return newly_flagged;
    };

    // ------------------------------------------------------------------ passes of the schedule (run_vslam.cpp:61-66) / the single call
    const int iters_early = iters;
    constexpr int npass = SCHED ? 3 : 1;
    bool done = false;
    int pass = 0;
    vslam_lm_stats* st = a.stats ? a.stats + w : nullptr;
    for (; pass < npass && !done; ++pass) {
        if (SCHED) { iters = pass < 2 ? iters_early : kRsSchedFinalIters; update_poses = pass == 2; }
        pass_init(pass == 0 ? 0 : 1);
        RPH(1);

        double lambda = 0, ni = 2, currentChi = 0, Bcur = 0;
        int it = 0, total_trials = 0;
        bool p_is_trial = false, need_backup = true, last_trial_is_current = true;
        currentChi = chi_pass(sm.Rt, Bcur);
        RPH(2);
        int bound = iters;
        for (;;) { // (at most twice: a continued pass re-enters the loop where it left it)
        for (; it < bound; ++it) {
            if (it == 0 && st && tid == 0) st->chi2_init = currentChi;
            double rho_gain = 0;
            int qmax = 0;
            bool again = true;
            while (again) {
                if (p_is_trial) { restore_positions(); p_is_trial = false; }
                // fixed-point scale of this state: every partial sum of every entry stays below Bcur (2 x margin in the exponent)
                int ex = 0;
                (void)frexp(2.0 * Bcur, &ex);
                const double scale = ldexp(1.0, 60 - ex), inv_scale = ldexp(1.0, ex - 60);
                bool ok2 = isfinite(Bcur);
                if (it == 0 && qmax == 0) { // computeLambdaInit: tau * max |H_jj| over every vertex
                    for (int i = tid; i < np; i += kRsBlock) sm.hdq[i] = 0;
                    if (tid == 0) sm.flag[6] = 0;
                    __syncthreads();
                    double md = linearise(0.0, scale, true);
                    __syncthreads();
                    if (tid < np) md = fmax(md, fabs((double)sm.hdq[tid] * inv_scale));
                    lambda = 1e-5 * block_max1(md);
                    ni = 2;
                    RPH(3);
                }
                for (int i = tid; i < nblk * 36; i += kRsBlock) Sq[i] = 0;
                for (int i = tid; i < np; i += kRsBlock) { sm.bpq[i] = 0; sm.bsq[i] = 0; }
                if (tid == 0) { sm.flag[6] = 0; sm.flag[10] = 0; }
                __syncthreads();
                linearise(lambda, scale, false);
                __syncthreads();
                RPH(4);
                // fixed point -> f64, lambda on the diagonal
                for (int i = tid; i < nblk * 36; i += kRsBlock) {
                    const int b = i / 36, e = i - 36 * b, rr = e / 6, cc = e - 6 * rr;
                    int I = 0;
                    while ((I + 1) * (I + 2) / 2 <= b) ++I;
                    const bool diagblk = (b - I * (I + 1) / 2) == I;
                    double v = (double)Sq[i] * inv_scale;
                    if (diagblk && rr == cc) v += lambda;
                    Sb[i] = v;
                }
                if (tid < np) { sm.bp[tid] = (double)sm.bpq[tid] * inv_scale; sm.bs[tid] = (double)sm.bsq[tid] * inv_scale; }
                __syncthreads();
                if (sm.flag[1]) ok2 = false;
                // Cholesky S = L L^T, right-looking over 6x6 block columns, the right-hand side as one more row (see lm_kernels.hip)
                for (int J = 0; J < nk && ok2; ++J) {
                    const int m = nk - J - 1, nrw = m * 6;
                    const bool rhs_row = tid == kRsBlock - 1;
                    if (tid < nrw || tid == 0 || rhs_row) {
                        double D[21];
                        const double* dj = &Sb[rs_blk(J, J)];
#pragma unroll
                        for (int i = 0; i < 6; ++i)
#pragma unroll
                            for (int j = 0; j <= i; ++j) D[i * (i + 1) / 2 + j] = dj[6 * i + j];
                        bool good = true;
                        double rd[6];
#pragma unroll
                        for (int j = 0; j < 6; ++j) {
                            double d = D[j * (j + 1) / 2 + j];
#pragma unroll
                            for (int kk = 0; kk < j; ++kk) d -= D[j * (j + 1) / 2 + kk] * D[j * (j + 1) / 2 + kk];
                            if (!(d > 0.0) || !isfinite(d)) good = false;
                            rd[j] = rsqrt_nr(d);
                            D[j * (j + 1) / 2 + j] = d * rd[j];
#pragma unroll
                            for (int i = j + 1; i < 6; ++i) {
                                double v = D[i * (i + 1) / 2 + j];
#pragma unroll
                                for (int kk = 0; kk < j; ++kk) v -= D[i * (i + 1) / 2 + kk] * D[j * (j + 1) / 2 + kk];
                                D[i * (i + 1) / 2 + j] = v * rd[j];
                            }
                        }
                        if (tid < nrw || rhs_row) {
                            double* rowv = rhs_row ? &sm.bs[6 * J] : &Sb[rs_blk(J + 1 + tid / 6, J) + 6 * (tid % 6)];
                            double x[6];
#pragma unroll
                            for (int c = 0; c < 6; ++c) {
                                double v = rowv[c];
#pragma unroll
                                for (int kk = 0; kk < c; ++kk) v -= x[kk] * D[c * (c + 1) / 2 + kk];
                                x[c] = v * rd[c];
                            }
#pragma unroll
                            for (int c = 0; c < 6; ++c) rowv[c] = x[c];
                        }
                        if (tid == 0) {
                            if (!good) sm.flag[1] = 1;
#pragma unroll
                            for (int i = 0; i < 21; ++i) sm.Ld[24 * J + i] = D[i];
#pragma unroll
                            for (int i = 0; i < 6; ++i) sm.rdiag[6 * J + i] = rd[i];
                        }
                    }
                    __syncthreads();
                    if (sm.flag[1]) { ok2 = false; break; }
                    const int npair = m * (m + 1) / 2, nitem = npair * 6 + m;
                    for (int t = tid; t < nitem; t += kRsBlock) {
                        const double* xi; double* out; int Kb;
                        if (t < npair * 6) {
                            const int pr = t / 6, rr = t - 6 * pr;
                            int aa = 0, rem = pr;
                            while (rem > aa) { rem -= aa + 1; ++aa; }
                            const int Ib = J + 1 + aa; Kb = J + 1 + rem;
                            xi = &Sb[rs_blk(Ib, J) + 6 * rr]; out = &Sb[rs_blk(Ib, Kb) + 6 * rr];
                        } else { Kb = J + 1 + (t - npair * 6); xi = &sm.bs[6 * J]; out = &sm.bs[6 * Kb]; }
                        const double2* xi2 = reinterpret_cast<const double2*>(xi);
                        const double2 a0 = xi2[0], a1 = xi2[1], a2 = xi2[2];
                        double2* o2 = reinterpret_cast<double2*>(out);
                        double2 o[3] = {o2[0], o2[1], o2[2]};
                        double v[6] = {o[0].x, o[0].y, o[1].x, o[1].y, o[2].x, o[2].y};
                        const double* kb = &Sb[rs_blk(Kb, J)];
#pragma unroll
                        for (int c = 0; c < 6; ++c) {
                            const double2* xk = reinterpret_cast<const double2*>(kb + 6 * c);
                            const double2 b0 = xk[0], b1 = xk[1], b2 = xk[2];
                            v[c] -= (a0.x * b0.x + a0.y * b0.y) + (a1.x * b1.x + a1.y * b1.y) + (a2.x * b2.x + a2.y * b2.y);
                        }
                        o2[0] = make_double2(v[0], v[1]); o2[1] = make_double2(v[2], v[3]); o2[2] = make_double2(v[4], v[5]);
                    }
                    __syncthreads();
                }
                if (sm.flag[1]) ok2 = false;
                __syncthreads();
                if (tid == 0) sm.flag[1] = 0;
                if (ok2) {
                    if (wave == 0) { // backward substitution L^T x = y, block by block from the last (see lm_kernels.hip)
                        double u0 = lane < np ? sm.bs[lane] : 0.0, u1 = lane + 64 < np ? sm.bs[lane + 64] : 0.0;
                        for (int J = nk - 1; J >= 0; --J) {
                            double Lr0[6], Lr1[6], Ld[21], rdj[6], t[6], x[6];
#pragma unroll
                            for (int c = 0; c < 6; ++c) {
                                Lr0[c] = lane < 6 * J ? Sb[rs_blk(J, lane / 6) + 6 * c + lane % 6] : 0.0;
                                Lr1[c] = (6 * J > 64 && lane + 64 < 6 * J) ? Sb[rs_blk(J, (lane + 64) / 6) + 6 * c + (lane + 64) % 6] : 0.0;
                                rdj[c] = sm.rdiag[6 * J + c];
                            }
#pragma unroll
                            for (int i = 0; i < 21; ++i) Ld[i] = sm.Ld[24 * J + i];
#pragma unroll
                            for (int c = 0; c < 6; ++c) {
                                const int rr = 6 * J + c;
                                t[c] = rr >= 64 ? readlane_f64(u1, rr - 64) : readlane_f64(u0, rr);
                            }
#pragma unroll
                            for (int c = 5; c >= 0; --c) {
                                double v = t[c];
#pragma unroll
                                for (int k = c + 1; k < 6; ++k) v -= Ld[k * (k + 1) / 2 + c] * x[k];
                                x[c] = v * rdj[c];
                            }
                            if (lane == 0)
#pragma unroll
                                for (int c = 0; c < 6; ++c) sm.xp[6 * J + c] = x[c];
#pragma unroll
                            for (int c = 0; c < 6; ++c) { u0 = fma(-Lr0[c], x[c], u0); u1 = fma(-Lr1[c], x[c], u1); }
                        }
                    }
                } else {
                    for (int i = tid; i < np; i += kRsBlock) sm.xp[i] = 0;
                }
                __syncthreads();
                RPH(5);
                if (tid < nk) {
                    double E[7];
                    se3::exp(&sm.xp[6 * tid], E);
                    se3::mul(E, &sm.T[7 * tid], &sm.TT[7 * tid]);
                    expand_pose(&sm.TT[7 * tid], &sm.RtT[12 * tid]);
                }
                __syncthreads();
                double scale_gain = 1.0, Btrial = 0;
                double tempChi = backsub_pass(lambda, need_backup, scale_gain, Btrial);
                need_backup = false;
                p_is_trial = true;
                last_trial_is_current = false;
                RPH(6);
                if (!ok2) tempChi = 1.7976931348623157e308;
                rho_gain = (currentChi - tempChi) / scale_gain;
                const bool accept = rho_gain > 0 && isfinite(tempChi);
                if (accept) {
                    double alpha = 1. - pow(2 * rho_gain - 1, 3);
                    alpha = fmin(alpha, 2. / 3.);
                    lambda *= fmax(1. / 3., alpha);
                    ni = 2;
                    currentChi = tempChi;
                    Bcur = Btrial;
                    __syncthreads();
                    for (int i = tid; i < nk * 7; i += kRsBlock) sm.T[i] = sm.TT[i];
                    for (int i = tid; i < nk * 12; i += kRsBlock) sm.Rt[i] = sm.RtT[i];
                    __syncthreads();
                    p_is_trial = false; need_backup = true; last_trial_is_current = true;
                } else {
                    lambda *= ni;
                    ni *= 2;
                }
                ++qmax;
                again = (rho_gain < 0) && qmax < 10;
            }
            total_trials += qmax;
            if (st && tid == 0 && it < VSLAM_LM_MAX_ITERS) { st->chi2_iter[it] = currentChi; st->lambda_iter[it] = lambda; st->trials_iter[it] = qmax; }
            if (qmax == 10 || rho_gain == 0) { ++it; bound = 0; break; }
        }
        if (st && tid == 0) { st->iterations = it; st->total_trials = total_trials; st->chi2_final = currentChi; st->lambda_final = lambda; }
        RPH(7);

        // ---- landmark write-back (:272-287): the ACCEPTED positions of this pass's graph (before the classification clears live words); while a
        // rejected trial sits in LDS the accepted state is the backup
        if (update_lms) {
            for (int s = tid; s < nl; s += kRsBlock)
                if (live[s] & 0xFFFu) {
                    const int l = perm[s];
                    double vx, vy, vz; // (the accepted state: the backup while a rejected trial sits in the working arrays)
                    if (p_is_trial) { vx = Pbak[s]; vy = Pbak[nl + s]; vz = Pbak[2 * (size_t)nl + s]; }
                    else if (s < sl0) { vx = Pw[s]; vy = Pw[nl + s]; vz = Pw[2 * (size_t)nl + s]; }
                    else { vx = P[s - sl0]; vy = P[nlpL + s - sl0]; vz = P[2 * nlpL + s - sl0]; }
                    a.xyz[3 * ((size_t)lm0 + l)] = (float)vx; a.xyz[3 * ((size_t)lm0 + l) + 1] = (float)vy; a.xyz[3 * ((size_t)lm0 + l) + 2] = (float)vz;
                }
        }
        // ---- chi2 of every active edge at the LAST EVALUATED state (g2o leaves the errors of the last trial behind, accepted or not), the
        // adaptive threshold and the landmark flags (optimization.cpp:224-266): the landmark's last edge decides
        const int newly_flagged = classify_pass(last_trial_is_current ? sm.Rt : sm.RtT);
        RPH(8);
        if (SCHED && pass < 2 && adaptive) {
            if (!done) { // (done: this was the continuation -- the last pass, whatever its own classification flagged)
                const bool repeatable = __syncthreads_or(newly_flagged) == 0; // (uniform) the next pass would see the inputs this one saw
                if (repeatable) {
                    done = true; update_poses = 1;
                    if (bound > 0 && bound < kRsSchedFinalIters) { bound = kRsSchedFinalIters; continue; } // continue it as the last pass
                }
            }
        }
        break;
        }
        // ---- write-back (:272-287)
        __syncthreads();
        if (update_poses) for (int i = tid; i < nk * 7; i += kRsBlock) a.T[Tbase + i] = sm.T[i];
        RPH(9);
    }
    if (SCHED && tid == 0) ra.passes[w] = pass;
    if (tid == 0) ra.status[w] = VSLAM_OK;
#undef RPH
}

// ------------------------------------------------------------------------------------------------------------- launcher
static int rs_lds_per_cu(int device) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, device) != hipSuccess || v <= 0) v = 64 * 1024;
    return v;
}
// dynamic LDS of the narrow form: two windows (workgroups of four waves at 256 VGPRs) share a CU, half of its LDS each
int rs_dyn_lds_bytes(int device) {
    const int avail = rs_lds_per_cu(device) / 2 - (int)sizeof(RsShared) - 256;
    return avail > 0 ? (avail & ~255) : 0;
}

int launch_ba_resident(const RsLaunch& L, hipStream_t stream) {
    RsArgs ra;
    memset(&ra, 0, sizeof(ra));
    ra.a = L.a;
    ra.uv_s = reinterpret_cast<float2*>(L.uv_s); ra.epos = L.epos; ra.tab = L.tab; ra.xin = L.xin; ra.Pbak = L.Pbak; ra.Dc = L.Dc; ra.blc = L.blc;
    ra.status = L.status; ra.passes = L.passes; ra.defer = L.defer; ra.order = L.order; ra.dbg = L.dbg;
    ra.want_chi2 = L.want_chi2; ra.dense_to_general = L.dense_to_general;
    // width: more windows than CUs -> 256 lanes, two windows per CU (throughput); otherwise 512 lanes and the whole CU's LDS (a window's latency).
    // Both widths give the same bits (see kRsStreams), so the choice may depend on the launch.
    static int s_cus[16] = {0}, s_full[16] = {0};
    int dev = 0;
    VS_HIP(hipGetDevice(&dev));
    const int di = dev >= 0 && dev < 16 ? dev : 0;
    if (!s_cus[di]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        const int full = rs_lds_per_cu(dev) - (int)sizeof(RsShared) - 256;
        s_full[di] = full > 0 ? (full & ~255) : 0;
        s_cus[di] = n;
    }
    const int lanes = L.lanes == 256 || L.lanes == 512 ? L.lanes : (L.a.n_windows <= s_cus[di] ? 512 : 256);
    const int dyn = lanes == 512 ? s_full[di] : L.dyn_bytes;
    ra.dyn_bytes = dyn; ra.dyn_narrow = L.dyn_bytes;
    if (!L.opt_in_done) { // more than 64 KB of dynamic LDS needs the opt-in (once per context, i.e. per device)
        VS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ba_resident_kernel<true, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, L.dyn_bytes));
        VS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ba_resident_kernel<false, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, L.dyn_bytes));
        VS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ba_resident_kernel<true, 512>), hipFuncAttributeMaxDynamicSharedMemorySize, s_full[di]));
        VS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&ba_resident_kernel<false, 512>), hipFuncAttributeMaxDynamicSharedMemorySize, s_full[di]));
    }
    if (lanes == 512) {
        if (L.schedule) hipLaunchKernelGGL((ba_resident_kernel<true, 512>), dim3(L.a.n_windows), dim3(512), (size_t)dyn, stream, ra, 5, 0, 0, 1, L.adaptive);
        else hipLaunchKernelGGL((ba_resident_kernel<false, 512>), dim3(L.a.n_windows), dim3(512), (size_t)dyn, stream, ra, L.iters, L.update_poses, L.update_lms, 1, 0);
    } else {
        if (L.schedule) hipLaunchKernelGGL((ba_resident_kernel<true, 256>), dim3(L.a.n_windows), dim3(256), (size_t)dyn, stream, ra, 5, 0, 0, 1, L.adaptive);
        else hipLaunchKernelGGL((ba_resident_kernel<false, 256>), dim3(L.a.n_windows), dim3(256), (size_t)dyn, stream, ra, L.iters, L.update_poses, L.update_lms, 1, 0);
    }
    VS_HIP(hipGetLastError());
    return VSLAM_OK;
}

} // namespace vslam
