// lm_kernels.hip -- K9..K12: Levenberg-Marquardt back-end on the GPU.
//
// Replaces the g2o optimiser inside optimize_map and optimize_pose_only
// (/root/reference/src/stereo_visual_slam_main/optimization.cpp:103-288, :290-436) and serves the north_star
// motion-only pose stage that stands in for cv::solvePnPRansac (visual_odometry.cpp:277).
//   residuals / Jacobians   : EdgeProjection (optimization.cpp:41-73), PoseOnlyEdgeProjection (:75-101)
//   vertex updates          : T <- exp(d) * T (:26-32), p <- p + d (:34-39)
//   optimiser               : g2o Levenberg (lambda0 = 1e-5 max diag H, accept: lambda *= max(1/3, min(2/3, 1-(2rho-1)^3)),
//                             reject: lambda *= ni, ni *= 2, <= 10 trials), Huber kernel (delta = 5.991), Schur
//                             complement on the landmarks, Cholesky on the reduced (6 n_kf)^2 system.
//
// gfx950 mapping: ONE WORKGROUP PER WINDOW runs the whole LM loop persistently (no host round trips; a batch of
// windows fills the 256 CUs).  All sums are f64 with a FIXED order (no floating-point atomics):
//   - evaluation + linearisation                : one keyframe-major pass per state: the edge lists are cut into 64-edge rows,
//                                                 every wave streams a contiguous range of rows through register queues,
//                                                 keeps the Huber weight of every edge and accumulates Hpp, b_p on the fly;
//                                                 trial states are linearised speculatively into spare buffers, so an
//                                                 accepted trial needs no re-evaluation (one call site: the initial state goes
//                                                 through the same code as a pseudo-iteration)
//   - per-landmark blocks (Hll, b_l, Dinv), back-substitution : one lane per landmark; records are recomputed from the
//                                                 landmark position and the observations (batched loads), not gathered;
//                                                 Dinv of the first kDinvLds landmarks stays in LDS
//   - Schur blocks S[k1][k2]                    : one wave per keyframe pair, dealt by work; off-diagonal pairs walk a
//                                                 precomputed 8-B hit list (landmark + two weights per hit, the camera-frame points
//                                                 are re-derived), diagonal pairs stream the keyframe's own list and
//                                                 also produce the reduced right-hand side; single owner per block
//   - wave reductions                           : halving butterfly (N sums cost ~N shuffles and end one per lane)
//   - reduced system                            : blocked left-looking Cholesky in LDS, the right-hand side rides along as an extra
//                                                 row (forward substitution for free), backward substitution with the unknowns
//                                                 in the lanes of one wave (readlane + FMA per unknown)
// Landmarks and pixels are f32 at rest (quirk Q4); poses, accumulators and the LM state are f64.
// No MFMA: the largest dense object is the 72x72 reduced system (and f64 MFMA has the vector rate on gfx950).  Bound: f64 VALU
// issue at two waves per SIMD (256 VGPRs: the Schur accumulators), the serial reduced-system phases and, first of all, the bytes
// streamed per iteration (256 windows x 1.5 MB do not fit the Infinity Cache; see DESIGN.md for the calibrated counters); machine facts used
// below (tools/scratch/lat.hip): one wave issues a VALU op per 8 cycles, a dependent f64 FMA takes 8, an LDS round trip ~60, a
// barrier of 8 waves ~210, and loops with run-time trip counts are NOT software-pipelined by the compiler -- every hot loop
// here fetches the operands of several iterations before the first use.  See DESIGN.md section 5 for the phase split.
#include "vslam_internal.h"

#include <stdlib.h>

#include <vector>

#include "se3_device.h"
#include "lm_device.h"

namespace vslam {

#ifndef VSLAM_LM_BLOCK
#define VSLAM_LM_BLOCK 512
#endif
#ifndef VSLAM_LM_DYNAMIC_ITEMS
#define VSLAM_LM_DYNAMIC_ITEMS 1
#endif
#ifndef VSLAM_LM_ITEM_FIXED
#define VSLAM_LM_ITEM_FIXED 128 // fixed cost of a Schur work item, in hits (see the item balance)
#endif
#ifndef VSLAM_LM_MIN_WAVES
#define VSLAM_LM_MIN_WAVES 2 // waves per SIMD the register allocation must leave room for
#endif
// VSLAM_LM_PRIO = n > 0 (tuning aid): the two waves of a SIMD (wave w and w + 4 of the 8-wave workgroup) alternate their issue
// priority every n rows of the hot loops, so that the older wave does not finish its share 25-30 % before its partner
#ifndef VSLAM_LM_PRIO
#define VSLAM_LM_PRIO 0
#endif
#if VSLAM_LM_PRIO > 0
#define LM_PRIO_TICK(cnt) do { if ((((cnt) / VSLAM_LM_PRIO) ^ (wave >> 2)) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); ++(cnt); } while (0)
#else
#define LM_PRIO_TICK(cnt) do {} while (0)
#endif
constexpr int kLmBlock = VSLAM_LM_BLOCK;
constexpr int kLmWaves = kLmBlock / 64;
constexpr int kMaxKf = VSLAM_MAX_KF;
constexpr int kMaxNp = 6 * kMaxKf;
constexpr int kMaxPairs = kMaxKf * (kMaxKf + 1) / 2;
constexpr int kHitsPerEdge = (kMaxKf + 1) / 2 + 1;
#ifndef VSLAM_LM_DINV_LDS
#define VSLAM_LM_DINV_LDS 1950
#endif
constexpr int kDinvLds = VSLAM_LM_DINV_LDS;  // landmarks whose Dinv stays in LDS (48 B each: the 96 KB the static state leaves free)
constexpr int kLin = 2;        // doubles per edge of linearisation scratch: the Huber weight at the current state / at the trial state
constexpr int kPoseParts = 3;  // a pose's by-pose edge list is summed by this many waves (parts added in a fixed order)
constexpr int kCntStride = 80; // per-wave counter row (>= kMaxPairs)
constexpr int kItemSlots = (kMaxPairs + kLmWaves - 1) / kLmWaves; // Schur work items (keyframe pairs) per wave
#ifndef VSLAM_LM_SLOTS
#define VSLAM_LM_SLOTS 5
#endif
constexpr int kDbgSlots = 24;               // phase cycle counters per window (VSLAM_LM_PROFILE)
constexpr int kLmSlots = VSLAM_LM_SLOTS;    // observations per landmark kept in the slot table (the rest is reached through the CSR)
constexpr int kSchedFinalIters = 10;        // iterations of the schedule's last optimize_map pass (run_vslam.cpp:66)
constexpr int kRowCap = 512;                // 64-landmark rows with a slot-width entry in LDS (32 768 landmarks per window)

size_t lm_hits_per_edge() { return kHitsPerEdge; }

struct alignas(16) LmShared {
    double S[kMaxNp * kMaxNp];
    double Hpp[kMaxKf * 36], HppT[kMaxKf * 36]; // pose blocks at the current state / at the state of the latest trial
    double bp[kMaxNp], bpT[kMaxNp], bs[kMaxNp], xp[kMaxNp];
    double rdiag[kMaxNp];                       // 1 / L_ii of the reduced system's Cholesky factor (the solves multiply)
    double Ld[kMaxKf * 24];                     // the factored diagonal blocks L_JJ (lower triangle, 21 of 24 slots each): read by the backward substitution only
    double Rt[kMaxKf * 12], RtTrial[kMaxKf * 12];
    double T[kMaxKf * 7], TTrial[kMaxKf * 7];
    double red[kLmWaves * 2];
    double part[kMaxKf * kPoseParts * 27]; // per (pose, part) partial sums of the pose blocks
    int ptot[kCntStride];                  // hits per keyframe pair
    int cnt[kLmWaves * kCntStride];        // per-wave counters / running offsets of the list builders
    uint8_t item[kLmWaves * kItemSlots];   // Schur work items (pair << 1 | row half) dealt to waves, 0xFF = none
    uint8_t pk1[kCntStride], pk2[kCntStride];
    int kfp[kMaxKf + 4];                   // kf_ptr (keyframe-major range starts), for the per-edge keyframe lookup
    int rowp[kMaxKf + 4];                  // first 64-edge row of every keyframe's list (rows never straddle keyframes)
    int flag[8];
    int pairp[kMaxPairs + 2];              // pair_ptr (first hit of every keyframe pair): read at every Schur item start -- from global memory that
                                           // was one dependent round trip per item before its first operand could even be requested
    uint8_t rowmax[kRowCap];               // most observations of an active landmark among landmarks [64 r, 64 r + 64): the landmark-wise phases fetch
                                           // slot q of a row only if q < rowmax[r] (rows beyond kRowCap: all slots)
};
static_assert(kMaxKf * kPoseParts >= kMaxKf + kLmWaves - 1, "part[] must hold one slot per (keyframe, wave) segment");
static_assert(kLmWaves <= 16, "cnt rows");
static_assert(kMaxNp <= 128, "the backward substitution keeps two unknowns per lane of a wave");
static_assert(kMaxPairs <= kLmWaves * kItemSlots, "item[] too small");

struct LmKernelArgs {
    LmWindowArgs a;
    // implicit single-pose problems (PnP): edge e <-> point e, keyframe 0
    const int32_t* pnp_n;
    int capacity;
    uint8_t* act;     // total_lm
    int32_t* kf_pos;  // total_edge
    double* chi2k;    // total_edge: chi2 per edge in keyframe-major order (scattered to the caller's order at the end)
    float* uvk;       // 2 x total_edge: observations in keyframe-major order
    int32_t* status;  // n_windows
    long long* dbg_cycles; // tuning aid (VSLAM_LM_PROFILE=1): 16 phase cycle counters per window, thread 0
    float* slot_uv;   // kLmSlots x total_lm float2: observation q of landmark l at [q * nl + l] (window-local), landmark-wise phases
    uint8_t* slot_kf; // kLmSlots x total_lm: its keyframe
    uint8_t* lcnt;    // total_lm: observations of an ACTIVE landmark, 0 = not in the graph
    int dinv_lds;     // landmarks per window whose Dinv is kept in dynamic LDS (0 = none)
    int want_chi2;    // the caller passed a chi2 output array: scatter chi2 back to its edge order at the end
    const int32_t* order; // n_windows: workgroup i takes window order[i] (largest first, lm_order_kernel), or null: window i
    int32_t* passes;      // n_windows: optimize_map passes of the current schedule that were EXECUTED for the window (written by the SCHED instance; 3 everywhere for the plain schedule)
    const int32_t* defer; // n_windows or null: with it, this launch only takes the windows ba_resident_kernel left (defer[w] != 0)
};

// slots of the landmark-wise slot table worth fetching for the 64-landmark row that starts at landmark `first` (wave-uniform)
__device__ inline int slot_width(const uint8_t* rowmax, int first) {
    const int row = first >> 6;
    return __builtin_amdgcn_readfirstlane(row < kRowCap ? (int)rowmax[min(row, kRowCap - 1)] : kLmSlots);
}

// deterministic block sum: per-thread partial -> wave butterfly -> waves summed in order
__device__ inline double block_sum(double v, double* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0;
    for (int w = 0; w < kLmWaves; ++w) s += red[w];
    return s;
}
__device__ inline double block_max(double v, double* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = red[0];
    for (int w = 1; w < kLmWaves; ++w) s = fmax(s, red[w]);
    return s;
}

// v - sum_{kk < 6 nK} a[kk] * b[kk] for two 16-B aligned LDS rows: the operands of block K + 1 are requested before the products
// of block K are formed, and the sum runs in two independent chains (a dependent f64 FMA issues every 8 cycles).
__device__ inline double row_dot_sub(double v, const double* a, const double* b, int nK) {
    const double2* ra2 = reinterpret_cast<const double2*>(a);
    const double2* rb2 = reinterpret_cast<const double2*>(b);
    if (nK <= 0) return v;
    double s0 = 0, s1 = 0;
    double2 a0 = ra2[0], a1 = ra2[1], a2 = ra2[2], b0 = rb2[0], b1 = rb2[1], b2 = rb2[2];
    for (int K = 1; K < nK; ++K) {
        const double2 c0 = ra2[3 * K], c1 = ra2[3 * K + 1], c2 = ra2[3 * K + 2];
        const double2 d0 = rb2[3 * K], d1 = rb2[3 * K + 1], d2 = rb2[3 * K + 2];
        s0 = fma(a0.x, b0.x, s0); s1 = fma(a0.y, b0.y, s1); s0 = fma(a1.x, b1.x, s0); s1 = fma(a1.y, b1.y, s1); s0 = fma(a2.x, b2.x, s0); s1 = fma(a2.y, b2.y, s1);
        a0 = c0; a1 = c1; a2 = c2; b0 = d0; b1 = d1; b2 = d2;
    }
    s0 = fma(a0.x, b0.x, s0); s1 = fma(a0.y, b0.y, s1); s0 = fma(a1.x, b1.x, s0); s1 = fma(a1.y, b1.y, s1); s0 = fma(a2.x, b2.x, s0); s1 = fma(a2.y, b2.y, s1);
    return v - (s0 + s1);
}

// SCHED: the in-kernel adaptive schedule (below).  A separate instance: the pass loop costs the single-pass code ~5 % in spilled registers, and the
// single-pass instance is what the host tier's per-call latency and the plain three-launch schedule run.
template <bool IMPL, bool SCHED = false>
__global__ __launch_bounds__(kLmBlock, VSLAM_LM_MIN_WAVES) void lm_window_kernel(LmKernelArgs ka, int mode, int iters, int update_poses, int update_lms,
                                                            int classify, int reuse_csr) {
    static_assert(!(IMPL && SCHED), "the schedule is a property of the window problems");
    const LmWindowArgs& a = ka.a;
    __shared__ LmShared sm;
    const int w = ka.order ? ka.order[blockIdx.x] : (int)blockIdx.x;
    if (ka.defer && !(ka.defer[w] & 1)) return; // (uniform) ba_resident_kernel has done this window
    int prio_cnt = 0; (void)prio_cnt;
    // keyframes of this window: a.n_kf slots (the pose stride), of which window w uses the first n_kf_w[w] (a growing map)
    const int nk = (!IMPL && a.n_kf_w) ? min(max(a.n_kf_w[w], 1), a.n_kf) : a.n_kf, np = 6 * nk;
    const size_t Tbase = (size_t)w * a.n_kf * 7;
    int lm0, nl, e0, ne;
    if (IMPL) { nl = min(max(ka.pnp_n[w], 0), ka.capacity); lm0 = w * ka.capacity; e0 = lm0; ne = nl; }
    else { lm0 = a.lm_off[w]; nl = a.lm_off[w + 1] - lm0; e0 = a.edge_off[w]; ne = a.edge_off[w + 1] - e0; }
    const float* xyz = a.xyz + 3 * (size_t)lm0;
    const float* uv = a.uv + 2 * (size_t)e0;
    const int32_t* kfi = IMPL ? nullptr : a.kf_idx + e0;
    const int32_t* lmi = IMPL ? nullptr : a.lm_idx + e0;
    double* P = a.P + 3 * (size_t)lm0;
    double* Pt = a.Ptrial + 3 * (size_t)lm0;
    double* Hll = a.Hll + 6 * (size_t)lm0;
    double* bl = a.bl + 3 * (size_t)lm0;
    double* Dinv = a.Dinv + 6 * (size_t)lm0;
    // Dinv of the first kDinvLds landmarks lives in LDS (written once per trial, gathered ~8 times per landmark by the Schur
    // passes and once by the back-substitution); the rest goes through global memory
    extern __shared__ double sDinv[];
    const int ncache = (!IMPL && mode == 0) ? min(nl, ka.dinv_lds) : 0;
    auto storeD = [&](int l, const double (&Di)[6]) {
        if (l < ncache) { double2* q = reinterpret_cast<double2*>(sDinv + 6 * l); q[0] = make_double2(Di[0], Di[1]); q[1] = make_double2(Di[2], Di[3]); q[2] = make_double2(Di[4], Di[5]); }
        else { double2* q = reinterpret_cast<double2*>(Dinv + 6 * (size_t)l); q[0] = make_double2(Di[0], Di[1]); q[1] = make_double2(Di[2], Di[3]); q[2] = make_double2(Di[4], Di[5]); }
    };
    // Prefetching form for the Schur loops: rows whose lanes are ALL cached (landmark ids ascend along a list, so that is every row but
    // one per list) read LDS on a wave-uniform path of their own.  On the mixed path the LDS reads and the global loads share their
    // destination registers and the compiler orders them with an `s_waitcnt vmcnt(0)` in front of the LDS reads -- right after the
    // row's prefetch loads were issued, i.e. one full memory round trip per row (650 of the 2500 cycles of a hit row).
    auto loadD_row = [&](int l, double2& Da, double2& Db, double2& Dc) {
        if (__ballot(l >= ncache) == 0ull) { const double2* q = reinterpret_cast<const double2*>(sDinv + 6 * l); Da = q[0]; Db = q[1]; Dc = q[2]; }
        else if (l < ncache) { const double2* q = reinterpret_cast<const double2*>(sDinv + 6 * l); Da = q[0]; Db = q[1]; Dc = q[2]; }
        else { const double2* q = reinterpret_cast<const double2*>(Dinv + 6 * (size_t)l); Da = q[0]; Db = q[1]; Dc = q[2]; }
    };
    auto loadD = [&](int l, double2& Da, double2& Db, double2& Dc) {
        if (l < ncache) { const double2* q = reinterpret_cast<const double2*>(sDinv + 6 * l); Da = q[0]; Db = q[1]; Dc = q[2]; }
        else { const double2* q = reinterpret_cast<const double2*>(Dinv + 6 * (size_t)l); Da = q[0]; Db = q[1]; Dc = q[2]; }
    };
    double* lin = a.lin + kLin * (size_t)e0;
    double* chi2 = a.chi2 + e0;
    double* chi2k = IMPL ? chi2 : ka.chi2k + e0;                       // keyframe-major (identity for the single-pose problem)
    float2* uvk2 = IMPL ? const_cast<float2*>(reinterpret_cast<const float2*>(uv)) : reinterpret_cast<float2*>(ka.uvk) + e0;
    int32_t* lm_ptr = a.lm_ptr + lm0 + w;
    int32_t* kf_ptr = a.kf_ptr + (size_t)w * (kMaxKf + 1);
    int32_t* kf_lm = a.kf_edges + e0;   // landmark of the edge stored at by-pose position j
    int32_t* kf_pos = ka.kf_pos + e0;    // by-pose (keyframe-major) position of edge e: where its linearisation record lives
    int32_t* pair_ptr = a.pair_ptr + (size_t)w * (kMaxPairs + 1);
    int2* hits = reinterpret_cast<int2*>(a.pair_hits) + (size_t)e0 * kHitsPerEdge; // off-diagonal pairs only: {pos1 | pos2 << 16, landmark}
    uint8_t* act = ka.act + lm0;
    uint8_t* lcnt = ka.lcnt + lm0;
    float2* suv = reinterpret_cast<float2*>(ka.slot_uv) + (size_t)kLmSlots * lm0; // [q * nl + l]
    uint8_t* skf = ka.slot_kf + (size_t)kLmSlots * lm0;
    const double K[4] = {a.K[0], a.K[1], a.K[2], a.K[3]};
    const CamK ck = make_camk(K);
    const double delta = a.huber_delta;
    const bool with_lm = (mode == 0);
    const int npairs = nk * (nk + 1) / 2;
    long long* cyc = ka.dbg_cycles ? ka.dbg_cycles + kDbgSlots * (size_t)w : nullptr;
    long long t_ph = cyc ? clock64() : 0;
    // (a fire-and-forget atomic: `cyc[i] += ...` made thread 0 wait a global round trip per marker, which the NEXT phase was then charged with)
#define PH(i) do { if (cyc && tid == 0) { const long long t1__ = clock64(); atomicAdd(reinterpret_cast<unsigned long long*>(cyc) + (i), (unsigned long long)(t1__ - t_ph)); t_ph = t1__; } } while (0)
    // component-major (SoA) scratch: consecutive lanes touch consecutive addresses in every edge- or landmark-ordered loop.
    // The only per-edge linearisation state that is STORED is the Huber weight (8 B, keyframe-major): the camera-frame point
    // is re-derived from the landmark wherever it is needed -- the kernel is bound by the bytes it streams, and a stored 32-B
    // record {X, Y, 1/Z, w} was read ~3.5 times per iteration.  Two sets: a trial evaluation records its weights into the
    // spare set; an accepted trial makes that set current, so the next iteration needs no re-evaluation.
    double* recW = lin;                    // Huber weight per edge, keyframe-major
    double* recW_alt = lin + (size_t)ne;
#define PC(ptr, c, l) (ptr)[(size_t)(c) * nl + (l)]

    // In-kernel ADAPTIVE schedule (SCHED; run_vslam.cpp:58-71 = optimize_map(5), optimize_map(5), optimize_map(10), the first two without
    // write-back).  Every pass starts from the SAME poses and landmarks; only the landmark flags carry over.  A pass whose classification flags
    // nothing new therefore leaves the next pass the inputs it had itself: the next 5-iteration pass would repeat it bit for bit (the kernel is
    // deterministic) and the 10-iteration pass would repeat its 5 iterations and then run 5 more.  Such a pass is simply CONTINUED to 10 iterations,
    // classified again and written back -- it IS the last pass -- and the window is done: 10 or 15 LM iterations instead of 20, results identical
    // to the plain schedule's (tests/test_gpu_lm.py::test_adaptive_schedule_is_bit_identical).  The passes of a window run back to back in ONE
    // launch: a window that is done frees its CU for the next window at once (three launches would wait for the slowest window three times).
    const int iters_early = iters, reuse_csr0 = reuse_csr;
    constexpr int npass = SCHED ? 3 : 1;
    bool done = false;       // (uniform) the schedule's last pass has been run, as a continued early pass
    int pass = 0;
    for (; pass < npass && !done; ++pass) {
    // The thread's own index is (re)derived INSIDE the pass loop, behind an opaque move in the schedule instance: everything computed from it (dozens of
    // per-thread addresses and lane constants of the LM loop) is then not loop-invariant, so the compiler cannot hoist it in front of the pass loop, keep it
    // alive across the register-hungry list builders and spill it there (47 spill stores at the loop head without this).
    int tid_l = threadIdx.x;
    if (SCHED) asm volatile("" : "+v"(tid_l));
    const int tid = tid_l, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6); // tell the compiler it is wave-uniform: wave-indexed control flow goes scalar
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const int slot27 = wave_slot<27>(lane), slot36 = wave_slot<36>(lane); // which butterfly sum this lane ends up holding
    if (SCHED) {
        iters = pass < 2 ? iters_early : kSchedFinalIters; update_poses = pass == 2; reuse_csr = reuse_csr0 || pass > 0;
        if (pass > 0) __syncthreads(); // (the classification of the previous pass wrote the flags this pass's setup reads)
    }
    // ------------------------------------------------------------------ setup
    if (reuse_csr0 && ka.status[w] != VSLAM_OK) return; // a later launch of the schedule: the first one rejected this window's indices
    if (tid < 8) sm.flag[tid] = 0;
    for (int i = tid; i < nk * 7; i += kLmBlock) sm.T[i] = a.T[Tbase + i];
    for (int i = tid; i < kLmWaves * kCntStride; i += kLmBlock) sm.cnt[i] = 0;
    if (tid < npairs) {
        int k1 = 0, rem = tid;
        while (rem >= nk - k1) { rem -= nk - k1; ++k1; }
        sm.pk1[tid] = (uint8_t)k1; sm.pk2[tid] = (uint8_t)(k1 + rem);
    }
    __syncthreads();
    if (tid < nk) expand_pose(&sm.T[7 * tid], &sm.Rt[12 * tid]);
    // (the setup loops read cold data: every loop issues the loads of several iterations before it consumes the first --
    // the compiler keeps a strided loop's iterations serial, one memory round trip each)
    for (int l0 = tid; l0 < nl; l0 += 3 * kLmBlock) {
        float v[3][3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int l = min(l0 + u * kLmBlock, nl - 1);
#pragma unroll
            for (int c = 0; c < 3; ++c) v[u][c] = xyz[3 * l + c];
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int l = l0 + u * kLmBlock;
            if (l < nl) { PC(P, 0, l) = (double)v[u][0]; PC(P, 1, l) = (double)v[u][1]; PC(P, 2, l) = (double)v[u][2]; }
        }
    }
    if (!IMPL && !reuse_csr) {
        // CSR by landmark from the sorted lm_idx; bad indices / unsorted input -> error flag
        for (int eb = tid; eb < ne; eb += 4 * kLmBlock) {
            int lv[4], lpv[4], kv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = min(eb + u * kLmBlock, ne - 1);
                lv[u] = lmi[e]; lpv[u] = e > 0 ? lmi[e - 1] : -1; kv[u] = kfi[e];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = eb + u * kLmBlock;
                if (e >= ne) break;
                const int l = lv[u], lp = lpv[u], k = kv[u];
                if (l < lp || l < 0 || l >= nl || k < 0 || k >= nk) { sm.flag[7] = 1; continue; }
                for (int x = lp + 1; x <= l; ++x) lm_ptr[x] = e;
                if (e == ne - 1) for (int x = l + 1; x <= nl; ++x) lm_ptr[x] = ne;
            }
        }
        if (ne == 0) for (int x = tid; x <= nl; x += kLmBlock) lm_ptr[x] = 0;
    }
    __syncthreads();
    if (!IMPL && ne > 0xFFFF && tid == 0) sm.flag[7] = 1; // Schur hit records pack two 16-bit record positions
    __syncthreads();
    if (sm.flag[7]) { // uniform
        if (tid == 0) { ka.status[w] = VSLAM_ERR_ARG; if (SCHED) ka.passes[w] = 0; }
        return;
    }
    for (int l0 = tid; l0 < nl; l0 += 3 * kLmBlock) {
        bool on[3];
        int cn[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int l = min(l0 + u * kLmBlock, nl - 1);
            on[u] = true; cn[u] = 1;
            if (!IMPL) { // (plain loads first, the logic after: a load behind `&&` sits in its own branch, one dependent round trip each)
                const int p0 = lm_ptr[l], p1 = lm_ptr[l + 1];
                const uint8_t inl = a.lm_inlier[lm0 + l];
                const uint8_t rel = (with_lm && a.reliable) ? a.reliable[lm0 + l] : (uint8_t)1;
                cn[u] = p1 - p0;
                on[u] = (cn[u] > 0) & (inl != 0) & (rel != 0);
            }
        }
#pragma unroll
        for (int u = 0; u < 3; ++u)
            if (l0 + u * kLmBlock < nl) { act[l0 + u * kLmBlock] = on[u]; if (!IMPL) lcnt[l0 + u * kLmBlock] = on[u] ? (uint8_t)min(cn[u], 255) : 0; }
        if (!IMPL) { // (a wave's lanes hold 64 consecutive, aligned landmarks per u: one row of the slot table)
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                int m = (l0 + u * kLmBlock < nl && on[u]) ? min(cn[u], 255) : 0;
                for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
                const int row = (l0 - lane + u * kLmBlock) >> 6;
                if (lane == 0 && row < kRowCap && l0 + u * kLmBlock < nl) sm.rowmax[row] = (uint8_t)m;
            }
        }
    }
    // Slot table of the landmark-wise phases: observation q < kLmSlots of landmark l at [q * nl + l].  A lane that owns landmark l
    // fetches its position, its count and its first observations in ONE round trip, coalesced across the lanes (through the CSR it
    // was lm_ptr -> edge list -> data: two dependent trips, the second one scattered).  The edges do not change inside a schedule.
    if (!IMPL && !reuse_csr && with_lm) {
        for (int eb = tid; eb < ne; eb += 4 * kLmBlock) {
            int lv[4], kv[4], bv[4]; float2 zv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int e = min(eb + u * kLmBlock, ne - 1); lv[u] = lmi[e]; kv[u] = kfi[e]; zv[u] = reinterpret_cast<const float2*>(uv)[e]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) bv[u] = lm_ptr[lv[u]];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = eb + u * kLmBlock, q = e - bv[u];
                if (e < ne && q < kLmSlots) { suv[(size_t)q * nl + lv[u]] = zv[u]; skf[(size_t)q * nl + lv[u]] = (uint8_t)kv[u]; }
            }
        }
    }
    __syncthreads();
    if (!IMPL) {
        PH(13);
        // ---- keyframe-major positions (ascending edge id inside a pose).  The per-edge weights, observations and landmark ids are
        // stored in this order: pose-wise phases stream them, Schur hits of a keyframe pair read two nearly contiguous runs, and
        // landmark-wise phases stay coalesced because neighbouring landmarks (creation order) sit at neighbouring positions.
        // Every wave owns a contiguous edge range, loads it once per pass and ranks its edges per keyframe with ballots.
        const int e_lo = (int)((long long)ne * wave / kLmWaves), e_hi = (int)((long long)ne * (wave + 1) / kLmWaves);
        for (int pass = 0; pass < 2; ++pass) {
            int run[kMaxKf];
#pragma unroll
            for (int kk = 0; kk < kMaxKf; ++kk) run[kk] = pass ? (int)sm.cnt[wave * kCntStride + kk] : 0;
            // landmark / keyframe ids are requested two batches ahead, the activity flag (a dependent gather) one batch ahead
            const int el = max(e_hi - 1, 0);
            int l1 = lmi[min(e_lo + lane, el)], l2 = lmi[min(e_lo + lane + 64, el)];
            int k1v = kfi[min(e_lo + lane, el)], k2v = kfi[min(e_lo + lane + 64, el)];
            int a1 = act[l1];
            const float2* uv2e = reinterpret_cast<const float2*>(uv);
            float2 z1 = uv2e[min(e_lo + lane, el)]; // (the observation travels with the ids: loaded inside the per-keyframe branches below it
                                                    // was one dependent round trip per keyframe present in the batch)
            for (int base = e_lo; base < e_hi; base += 64) {
                const int e = base + lane;
                const int l3 = lmi[min(e + 128, el)], k3v = kfi[min(e + 128, el)];
                const int a2 = act[l2];
                const float2 z2 = uv2e[min(e + 64, el)];
                const float2 zcur = z1;
                z1 = z2;
                const int k = (e < e_hi && a1) ? k1v : -1;
                const int lcur = l1;
                l1 = l2; l2 = l3; k1v = k2v; k2v = k3v; a1 = a2;
#pragma unroll
                for (int kk = 0; kk < kMaxKf; ++kk) {
                    if (kk < nk) {
                        const unsigned long long m = __ballot(k == kk);
                        if (pass && k == kk) { const int slot = run[kk] + __popcll(m & lt_mask); kf_pos[e] = slot; kf_lm[slot] = lcur; uvk2[slot] = zcur; }
                        run[kk] += __popcll(m);
                    }
                }
            }
            if (!pass) {
                if (lane == 0)
#pragma unroll
                    for (int kk = 0; kk < kMaxKf; ++kk) sm.cnt[wave * kCntStride + kk] = run[kk];
                __syncthreads();
                if (tid == 0) {
                    int acc = 0;
                    for (int kk = 0; kk < nk; ++kk) {
                        kf_ptr[kk] = acc;
                        for (int ww = 0; ww < kLmWaves; ++ww) { const int c = sm.cnt[ww * kCntStride + kk]; sm.cnt[ww * kCntStride + kk] = acc; acc += c; }
                    }
                    kf_ptr[nk] = acc;
                }
                __syncthreads();
                if (tid <= nk) sm.kfp[tid] = kf_ptr[tid];
            }
        }
        __syncthreads();
        if (with_lm) {
            PH(14);
            // ---- Schur hit lists per off-diagonal keyframe pair, landmark-ordered: wave = contiguous landmark range, 2 passes
            // (count, then write).  A lane packs the 16-bit record position of its landmark in every keyframe into six words
            // (0xFFFF = not observed); the pair loop is unrolled over compile-time (k1, k2), so the membership tests are register
            // bit-field compares and the per-pair running offsets stay in registers.
            for (int i = tid; i < kLmWaves * kCntStride; i += kLmBlock) sm.cnt[i] = 0;
            __syncthreads();
            const int l_lo = (int)((long long)nl * wave / kLmWaves), l_hi = (int)((long long)nl * (wave + 1) / kLmWaves);
            for (int pass = 0; pass < 2; ++pass) {
                int run[kMaxKf * (kMaxKf - 1) / 2];
                {
                    int idx = 0;
#pragma unroll
                    for (int K1 = 0; K1 < kMaxKf - 1; ++K1)
#pragma unroll
                        for (int K2 = K1 + 1; K2 < kMaxKf; ++K2) {
                            run[idx] = (pass && K2 < nk) ? (int)sm.cnt[wave * kCntStride + K1 * nk - K1 * (K1 - 1) / 2 + (K2 - K1)] : 0;
                            ++idx;
                        }
                }
                for (int base = l_lo; base < l_hi; base += 64) {
                    const int l = base + lane;
                    uint32_t w6[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
                    // (two levels of unconditional, clamped loads -- flag and list bounds, then the first five observations -- instead of a
                    // chain inside the branch: flag -> bounds -> observations was three dependent round trips per row of 64 landmarks)
                    const int lc = min(l, max(nl - 1, 0));
                    const uint8_t act_l = act[lc];
                    const int b0 = lm_ptr[lc], b1 = lm_ptr[lc + 1];
                    constexpr int kPre = 5;
                    int kq[kPre], pq[kPre];
#pragma unroll
                    for (int q = 0; q < kPre; ++q) { const int e = min(max(b0 + q, 0), max(ne - 1, 0)); kq[q] = kfi[e]; pq[q] = kf_pos[e]; }
                    if (l < l_hi && act_l) {
                        auto put = [&](int k, int ps) {
                            const int wi = k >> 1, sh = 16 * (k & 1);
#pragma unroll
                            for (int i = 0; i < 6; ++i)
                                if (i == wi) {
                                    if (((w6[i] >> sh) & 0xFFFFu) != 0xFFFFu) sm.flag[7] = 2; // duplicate (keyframe, landmark) edge
                                    w6[i] = (w6[i] & ~(0xFFFFu << sh)) | ((uint32_t)ps << sh);
                                }
                        };
#pragma unroll
                        for (int q = 0; q < kPre; ++q) if (b0 + q < b1) put(kq[q], pq[q]);
                        for (int e = b0 + kPre; e < b1; ++e) put(kfi[e], kf_pos[e]);
                    }
                    // which lanes' landmarks are seen by keyframe K: one ballot per keyframe, then a pair's hit mask is the AND of
                    // two scalar masks (creation-ordered landmarks: most pairs of a row are empty and cost three scalar ops)
                    unsigned long long pres[kMaxKf];
#pragma unroll
                    for (int K = 0; K < kMaxKf; ++K) pres[K] = K < nk ? __ballot(((w6[K >> 1] >> (16 * (K & 1))) & 0xFFFFu) != 0xFFFFu) : 0ull;
                    int idx = 0;
#pragma unroll
                    for (int K1 = 0; K1 < kMaxKf - 1; ++K1)
#pragma unroll
                        for (int K2 = K1 + 1; K2 < kMaxKf; ++K2) {
                            const unsigned long long m = pres[K1] & pres[K2];
                            if (m) { // uniform
                                if (pass && ((m >> lane) & 1ull)) {
                                    const uint32_t f1 = (w6[K1 >> 1] >> (16 * (K1 & 1))) & 0xFFFFu, f2 = (w6[K2 >> 1] >> (16 * (K2 & 1))) & 0xFFFFu;
                                    hits[run[idx] + __popcll(m & lt_mask)] = make_int2((int)(f1 | (f2 << 16)), l);
                                }
                                run[idx] += __popcll(m);
                            }
                            ++idx;
                        }
                }
                if (!pass && lane == 0) {
                    int idx = 0;
#pragma unroll
                    for (int K1 = 0; K1 < kMaxKf - 1; ++K1)
#pragma unroll
                        for (int K2 = K1 + 1; K2 < kMaxKf; ++K2) {
                            if (K2 < nk) sm.cnt[wave * kCntStride + K1 * nk - K1 * (K1 - 1) / 2 + (K2 - K1)] = run[idx];
                            ++idx;
                        }
                }
                if (!pass) {
                    __syncthreads();
                    if (tid < npairs) { int t = 0; for (int ww = 0; ww < kLmWaves; ++ww) t += sm.cnt[ww * kCntStride + tid]; sm.ptot[tid] = t; }
                    __syncthreads();
                    if (tid == 0) { int acc = 0; for (int p = 0; p < npairs; ++p) { pair_ptr[p] = acc; sm.pairp[p] = acc; acc += sm.ptot[p]; } pair_ptr[npairs] = acc; sm.pairp[npairs] = acc; }
                    __syncthreads();
                    if (tid < npairs) { int run = pair_ptr[tid]; for (int ww = 0; ww < kLmWaves; ++ww) { const int c = sm.cnt[ww * kCntStride + tid]; sm.cnt[ww * kCntStride + tid] = run; run += c; } }
                    __syncthreads();
                }
            }
            if (sm.flag[7]) { if (tid == 0) { ka.status[w] = VSLAM_ERR_ARG; if (SCHED) ka.passes[w] = 0; } return; } // (uniform: read after the barriers of the list build)
            // ---- balance the Schur work: item = keyframe pair; rank by hit count, deal ranks to waves in snake order
            const int nitems = npairs;
            for (int i = tid; i < kLmWaves * kItemSlots; i += kLmBlock) sm.item[i] = 0xFF;
            __syncthreads();
            // (work per item staged in LDS first: ranking straight from kf_ptr costs a global round trip per comparison)
            int* s_work = sm.cnt; // free after the list build
            // (cost of an item in hit-equivalents: its rows of 64 hits plus a fixed part -- the first-row round trips and the 36-value butterfly
            // are worth about two rows; with the short lists of a real sequence the fixed part is most of an item)
            if (tid < nitems) {
                const int a1 = sm.pk1[tid];
                const int hits_i = a1 == sm.pk2[tid] ? sm.kfp[a1 + 1] - sm.kfp[a1] : (int)sm.ptot[tid];
                s_work[tid] = hits_i > 0 ? ((hits_i + 63) & ~63) + VSLAM_LM_ITEM_FIXED : 0;
            }
            __syncthreads();
            if (tid < nitems) {
                const int mine = s_work[tid];
                int rank = 0;
                for (int j = 0; j < nitems; ++j) { const int c = s_work[j]; rank += (c > mine) || (c == mine && j < tid); }
#if VSLAM_LM_DYNAMIC_ITEMS
                sm.item[rank] = (uint8_t)tid; // costliest first: the waves draw the items from this list as they become free (below)
#else
                const int row = rank / kLmWaves, col = rank % kLmWaves;
                sm.item[((row & 1) ? kLmWaves - 1 - col : col) * kItemSlots + row] = (uint8_t)tid;
#endif
            }
        }
    }
    if (IMPL && tid == 0) { sm.kfp[0] = 0; sm.kfp[1] = ne; }
    __syncthreads();
    if (tid == 0) { // the evaluation pass walks the keyframe-major lists in rows of 64 edges
        int r = 0;
        for (int kk = 0; kk < nk; ++kk) { sm.rowp[kk] = r; r += (sm.kfp[kk + 1] - sm.kfp[kk] + 63) >> 6; }
        sm.rowp[nk] = r;
    }
    __syncthreads();

    PH(0);
    // ------------------------------------------------------------------ LM iterations
    double lambda = 0, ni = 2, currentChi = 0;
    int it = 0, total_trials = 0;
    vslam_lm_stats* st = a.stats ? a.stats + w : nullptr;

    // Every phase below is a chain of dependent gathers out of a per-window working set that lives in HBM (1.5 MB x
    // hundreds of windows), so each loop is written as batches: all loads of one dependency level for kEvalU items first,
    // then the arithmetic -- the per-thread summation order is unchanged.
    const float2* uv2 = reinterpret_cast<const float2*>(uv);
    constexpr int kEvalU = 4;
#ifndef VSLAM_LM_U
#define VSLAM_LM_U 2
#endif
    constexpr int kLmU = VSLAM_LM_U, kLmE = kLmSlots; // landmark-wise phases: landmarks per batch, observations preloaded per landmark
    // Evaluation + linearisation at (Rt, Pcur) in one keyframe-major pass.  The keyframe-major edge lists are cut into rows
    // of 64 edges (a row never straddles two keyframes); every wave owns a contiguous range of rows and streams it through
    // a register queue (landmark ids and observations two rows ahead, landmark positions one row ahead -- a row is ~2000 cycles of issue, deeper queues only cost register moves: the lists come from
    // HBM / Infinity Cache at ~1 us per dependent access, a row is ~0.5 us of arithmetic).  It writes the Huber
    // weight of every edge for the Schur phase and accumulates the pose blocks (H_pp upper triangle, b_p) on the fly, so the errors
    // never have to be stored; when the keyframe changes the 27 sums are folded (butterfly) into slot (keyframe + wave) --
    // unique, because the waves' row ranges are ordered like the keyframes.  Returns the robust chi2.  Trial states go
    // through the same pass into the spare buffers: an accepted trial is already linearised.
    const int ntot = sm.kfp[nk]; // active edges
    auto LMJ = [&](int j) -> int { return IMPL ? j : kf_lm[j]; };
    struct RowIt { int k, j, jend; };
    auto row_next = [&](RowIt& r) {
        r.j += 64;
        while (r.j >= r.jend && r.k + 1 < nk) { ++r.k; r.j = sm.kfp[r.k]; r.jend = sm.kfp[r.k + 1]; }
    };
    const int nrows = sm.rowp[nk], rows_per_wave = (nrows + kLmWaves - 1) / kLmWaves;
    auto eval = [&](const double* Rt, const double* Pcur, double* dstW, double* Hdst, double* bdst) -> double {
        double part = 0;
        const long long t_ev = cyc ? clock64() : 0;
        const int ra = min(wave * rows_per_wave, nrows), rb = min(ra + rows_per_wave, nrows);
        if (ra < rb) {
            RowIt cur;
            {
                int k = 0;
                while (k + 1 < nk && sm.rowp[k + 1] <= ra) ++k;
                cur.k = k; cur.j = sm.kfp[k] + 64 * (ra - sm.rowp[k]); cur.jend = sm.kfp[k + 1];
            }
            RowIt ahead = cur;
#ifndef VSLAM_LM_EVAL_PD
#define VSLAM_LM_EVAL_PD 1
#endif
#ifndef VSLAM_LM_EVAL_QD
#define VSLAM_LM_EVAL_QD 2
#endif
            constexpr int kPD = VSLAM_LM_EVAL_PD; // rows of landmark positions in flight
            constexpr int kQD = VSLAM_LM_EVAL_QD; // rows of landmark ids / observations in flight (>= kPD)
            static_assert(kQD >= kPD && kPD >= 1, "queue depths");
            int lq[kQD];
            float2 zq[kQD];
            double pq[kPD][3];
#pragma unroll
            for (int q = 0; q < kQD; ++q) {
                const int jj = min(ahead.j + lane, ahead.jend - 1);
                lq[q] = LMJ(jj); zq[q] = uvk2[jj];
                row_next(ahead);
            }
#pragma unroll
            for (int q = 0; q < kPD; ++q)
#pragma unroll
                for (int c = 0; c < 3; ++c) pq[q][c] = PC(Pcur, c, lq[q]);
            double acc[27];
#pragma unroll
            for (int i = 0; i < 27; ++i) acc[i] = 0;
            int kacc = cur.k;
            double Rk[12]; // pose of the current keyframe, wave-uniform
#pragma unroll
            for (int i = 0; i < 12; ++i) Rk[i] = uniform_f64(Rt[12 * kacc + i]);
            auto flush = [&](int k) {
                wave_reduce_scatter<27>(acc, lane);
                if (slot27 >= 0) sm.part[(k + wave) * 27 + slot27] = acc[0];
#pragma unroll
                for (int i = 0; i < 27; ++i) acc[i] = 0;
            };
            // (not unrolled: one copy of the row body and of the butterfly keeps the loop inside the instruction cache; the
            // queues are rotated with register moves instead of static indices)
#pragma unroll 1
            for (int row = ra;; ++row) {
                LM_PRIO_TICK(prio_cnt);
                const bool done = row >= rb;
                if (done || cur.k != kacc) {
                    flush(kacc);
                    if (done) break;
                    kacc = cur.k;
#pragma unroll
                    for (int i = 0; i < 12; ++i) Rk[i] = uniform_f64(Rt[12 * kacc + i]);
                }
                const float2 z1 = zq[0];
                const double px = pq[0][0], py = pq[0][1], pz = pq[0][2];
                { // rotate the queues and refill their tails: ids of row + kQD, positions of row + kPD
                    const int jj = min(ahead.j + lane, ahead.jend - 1);
#pragma unroll
                    for (int q = 0; q + 1 < kQD; ++q) { lq[q] = lq[q + 1]; zq[q] = zq[q + 1]; }
                    lq[kQD - 1] = LMJ(jj); zq[kQD - 1] = uvk2[jj];
                    row_next(ahead);
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
#pragma unroll
                        for (int q = 0; q + 1 < kPD; ++q) pq[q][c] = pq[q + 1][c];
                        pq[kPD - 1][c] = PC(Pcur, c, lq[kPD - 1]);
                    }
                }
                const int j = cur.j + lane;
                if (j < cur.jend) {
                    double x, y, ri, wgt, ex, ey, c, rho, A[12], wA[12];
                    cam_norm(Rk, px, py, pz, x, y, ri);
                    eval_obs(ck, x, y, z1, delta, ex, ey, c, rho, wgt); // (ex, ey: normalised error)
                    part += rho;
                    if (with_lm) dstW[j] = wgt; // the Schur passes re-derive x, y, 1/Z from the landmark; only the weight is kept
                    jac_norm(x, y, ri, A);
                    const double l0 = wgt * ck.fx2, l1 = wgt * ck.fy2;
#pragma unroll
                    for (int i = 0; i < 6; ++i) { wA[i] = l0 * A[i]; wA[6 + i] = l1 * A[6 + i]; }
                    int idx = 0;
#pragma unroll
                    for (int r = 0; r < 6; ++r)
#pragma unroll
                        for (int cc = r; cc < 6; ++cc) { acc[idx] = a_fma_pair(wA, r, A, cc, acc[idx]); ++idx; }
#pragma unroll
                    for (int r = 0; r < 6; ++r) acc[21 + r] = a_fma2(wA, r, -ex, -ey, acc[21 + r]);
                }
                row_next(cur);
            }
        }
        const double total = block_sum(part, sm.red); // (its barriers also publish sm.part)
        for (int t = tid; t < nk * 27; t += kLmBlock) { // the segments of a keyframe summed in wave order
            const int k = t / 27, i = t - k * 27;
            const int r0 = sm.rowp[k], r1 = sm.rowp[k + 1];
            double v = 0;
            for (int ww = 0; ww < kLmWaves; ++ww) {
                const int wa = min(ww * rows_per_wave, nrows), wb = min(wa + rows_per_wave, nrows);
                if (max(wa, r0) < min(wb, r1)) v += sm.part[(k + ww) * 27 + i];
            }
            if (i < 21) {
                int r = 0, rem = i;
                while (rem >= 6 - r) { rem -= 6 - r; ++r; }
                const int c = r + rem;
                Hdst[36 * k + 6 * r + c] = v; Hdst[36 * k + 6 * c + r] = v;
            } else bdst[6 * k + i - 21] = v;
        }
        __syncthreads();
        return total;
    };
    // chi2 of every active edge at (Rt, Pcur), keyframe-major: only needed once, for the classification / the caller
    auto chi_pass = [&](const double* Rt, const double* Pcur) {
        for (int base = tid; base < ntot; base += kEvalU * kLmBlock) {
            int l[kEvalU], k[kEvalU];
            float2 z[kEvalU];
#pragma unroll
            for (int u = 0; u < kEvalU; ++u) {
                const int j = min(base + u * kLmBlock, ntot - 1);
                l[u] = LMJ(j); z[u] = uvk2[j];
                int kk = 0;
                if (!IMPL)
                    for (int q = 1; q < nk; ++q) kk += j >= sm.kfp[q];
                k[u] = kk;
            }
            double px[kEvalU], py[kEvalU], pz[kEvalU];
#pragma unroll
            for (int u = 0; u < kEvalU; ++u) { px[u] = PC(Pcur, 0, l[u]); py[u] = PC(Pcur, 1, l[u]); pz[u] = PC(Pcur, 2, l[u]); }
#pragma unroll
            for (int u = 0; u < kEvalU; ++u) {
                const int j = base + u * kLmBlock;
                if (j >= ntot) continue;
                double X, Y, Z, ex, ey;
                project_err(&Rt[12 * k[u]], K, px[u], py[u], pz[u], z[u].x, z[u].y, X, Y, Z, ex, ey);
                chi2k[j] = ex * ex + ey * ey;
            }
        }
        __syncthreads();
    };
    bool last_trial_is_current = true; // g2o leaves the edge errors of the LAST EVALUATED trial behind (accepted or not)

    // The evaluation pass is the largest piece of code in the loop and the loop body has to stay inside the 64 KB instruction
    // cache two CUs share, so there is ONE call site: the initial state is evaluated by a pseudo-iteration (it = -1, "boot")
    // that skips the solve and runs the trial evaluation on the current buffers.  Every later iteration starts from an
    // accepted trial, which is already evaluated and linearised.
    int bound = iters;       // iterations of this pass; raised to kSchedFinalIters when an adaptive pass is continued; 0 once the loop has ended by
                             // g2o's own rule (ten failed trials / zero gain): more iterations would not run either
    it = iters > 0 ? -1 : 0;
    for (;;) { // (at most twice: a continued pass re-enters the loop where it left it)
    for (; it < bound; ++it) {
        const bool boot = it < 0;
        PH(1);
        if (it == 0 && st && tid == 0) st->chi2_init = currentChi;
        // ---- trial loop.  Every trial starts with the landmark pass (blocks H_ll, b_l and, once lambda is known, D^-1 straight
        // from the registers): H_ll is never stored except on the very first trial, whose lambda comes out of the pass itself.
        // A rejected trial therefore repeats the pass with the new lambda instead of re-reading 48 B per landmark -- rejections
        // are rare, the 144 KB of H_ll per iteration were 40 % of the kernel's write traffic.
        double rho_gain = 0;
        int qmax = 0;
        bool again = true;
        while (again) {
        const bool lam_known = it > 0 || qmax > 0;
        // ---- buildSystem: landmark blocks
        double maxdiag = 0;
        if (with_lm && !boot) {
            for (int l0 = tid; l0 < nl; l0 += kLmU * kLmBlock) {
                int cn[kLmU];
                double px[kLmU], py[kLmU], pz[kLmU];
                bool on[kLmU];
                int kk[kLmU][kLmE]; float2 zz[kLmU][kLmE];
                // one round trip: count, position and the first kLmE observations of the lane's landmarks (slot table)
#pragma unroll
                for (int u = 0; u < kLmU; ++u) {
                    const int l = min(l0 + u * kLmBlock, nl - 1);
                    cn[u] = lcnt[l];
                    px[u] = PC(P, 0, l); py[u] = PC(P, 1, l); pz[u] = PC(P, 2, l);
                    const int rm = slot_width(sm.rowmax, l0 - lane + u * kLmBlock); // (uniform)
                    // (slots nobody in the row has are redirected to slot 0, whose lines this batch fetches anyway: plain loads, all in flight
                    // together -- a branch around them cost 3-8 % (the compiler waits per branch) -- and no line of the unused slots is touched)
#pragma unroll
                    for (int q = 0; q < kLmE; ++q) { const size_t qo = (size_t)(q < rm ? q : 0) * nl + l; kk[u][q] = skf[qo]; zz[u][q] = suv[qo]; }
                }
#pragma unroll
                for (int u = 0; u < kLmU; ++u) on[u] = cn[u] > 0 && l0 + u * kLmBlock < nl;
#pragma unroll
                for (int u = 0; u < kLmU; ++u) {
                    if (!on[u]) continue;
                    const int l = l0 + u * kLmBlock;
                    double h[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
                    auto add_edge = [&](int k, float2 z) {
                        double x, y, ri, wg, ex, ey, c, rho, B[6], Bs[6];
                        cam_norm(&sm.Rt[12 * k], px[u], py[u], pz[u], x, y, ri);
                        eval_obs(ck, x, y, z, delta, ex, ey, c, rho, wg);
                        jac_point_norm(x, y, ri, &sm.Rt[12 * k], B);
                        const double l0 = wg * ck.fx2, l1 = wg * ck.fy2;
#pragma unroll
                        for (int i = 0; i < 3; ++i) { Bs[i] = l0 * B[i]; Bs[3 + i] = l1 * B[3 + i]; }
                        h[0] += Bs[0] * B[0] + Bs[3] * B[3]; h[1] += Bs[0] * B[1] + Bs[3] * B[4]; h[2] += Bs[0] * B[2] + Bs[3] * B[5];
                        h[3] += Bs[1] * B[1] + Bs[4] * B[4]; h[4] += Bs[1] * B[2] + Bs[4] * B[5]; h[5] += Bs[2] * B[2] + Bs[5] * B[5];
                        g[0] -= Bs[0] * ex + Bs[3] * ey; g[1] -= Bs[1] * ex + Bs[4] * ey; g[2] -= Bs[2] * ex + Bs[5] * ey;
                    };
#pragma unroll
                    for (int q = 0; q < kLmE; ++q) if (q < cn[u]) add_edge(kk[u][q], zz[u][q]);
                    if (cn[u] > kLmE) { // more than kLmE observations (long tracks of a real sequence): the rest through the CSR, ALL of them
                        // requested at once -- one edge per loop trip was one dependent round trip per observation, up to seven per batch
                        constexpr int kTail = kMaxKf - kLmE;
                        const int b0 = lm_ptr[l];
                        int kt[kTail]; float2 zt[kTail];
#pragma unroll
                        for (int q = 0; q < kTail; ++q) { const int e = min(b0 + kLmE + q, max(ne - 1, 0)); kt[q] = kfi[e]; zt[q] = uv2[e]; }
#pragma unroll
                        for (int q = 0; q < kTail; ++q) if (kLmE + q < cn[u]) add_edge(kt[q], zt[q]);
                        for (int e = b0 + kMaxKf; e < b0 + cn[u]; ++e) add_edge(kfi[e], uv2[e]); // (only a malformed graph gets here; it is rejected by the hit-list build)
                    }
                    if (!lam_known) {
#pragma unroll
                        for (int i = 0; i < 6; ++i) PC(Hll, i, l) = h[i];
                    }
#pragma unroll
                    for (int i = 0; i < 3; ++i) PC(bl, i, l) = g[i];
                    maxdiag = fmax(maxdiag, fmax(fabs(h[0]), fmax(fabs(h[3]), fabs(h[5]))));
                    if (lam_known) { // invert here: no pass over stored blocks
                        double Di[6];
                        if (!inv3_sym(h[0] + lambda, h[1], h[2], h[3] + lambda, h[4], h[5] + lambda, Di)) sm.flag[1] = 1;
                        storeD(l, Di);
                    }
                }
            }
        }
        PH(2);
        PH(3);
        if (it == 0 && qmax == 0) { // computeLambdaInit: tau * max |H_jj| over every vertex
            if (tid < np) maxdiag = fmax(maxdiag, fabs(sm.Hpp[36 * (tid / 6) + 7 * (tid % 6)]));
            lambda = 1e-5 * block_max(maxdiag, sm.red);
            ni = 2;
        }
        {
            bool ok2 = true;
            double scale = 1.0;
            if (!boot) {
            if (with_lm) {
                // Dinv = (Hll + lambda I)^-1 per landmark on the first trial of a call (every other trial got it from the landmark pass)
                if (!lam_known) {
                    int bad = 0;
                    for (int l = tid; l < nl; l += kLmBlock) {
                        if (!act[l]) continue;
                        double Di[6];
                        if (!inv3_sym(PC(Hll, 0, l) + lambda, PC(Hll, 1, l), PC(Hll, 2, l), PC(Hll, 3, l) + lambda, PC(Hll, 4, l), PC(Hll, 5, l) + lambda, Di)) bad = 1;
                        storeD(l, Di);
                    }
                    if (bad) sm.flag[1] = 1;
                }
                for (int i = tid; i < np * np; i += kLmBlock) sm.S[i] = 0;
                if (tid == 0) sm.flag[6] = 0; // next Schur item to hand out
                __syncthreads(); // Dinv visible (global, same workgroup) + S zeroed
                PH(4);
                PH(5);
                // Schur blocks: S[k1][k2] = [k1==k2](Hpp + lambda I) - sum_hits W1 Dinv W2^T.  item = (pair, row half), owned by one wave
                // Items are handed out DYNAMICALLY, costliest first: a wave that is done draws the next one from an LDS counter.  (A static deal by
                // estimated cost left the waves 22-27 % of the pass waiting for the slowest one: the older wave of a SIMD runs faster than its
                // partner, and an item's fixed cost is not proportional to its hits.)  Which wave computes an item does not change its result: one
                // wave owns a block and reduces it in a fixed order.
#if VSLAM_LM_DYNAMIC_ITEMS
                for (;;) {
                    int slot = 0;
                    if (lane == 0) slot = atomicAdd(&sm.flag[6], 1);
                    slot = __builtin_amdgcn_readfirstlane(slot);
                    if (slot >= npairs) break;
                    const int p = sm.item[slot];
#else
                for (int slot = 0; slot < kItemSlots; ++slot) {
                    const int p = sm.item[wave * kItemSlots + slot];
#endif
                    if (p == 0xFF) continue; // uniform per wave
                    PH(23); // (item boundary: what came before was the previous item's reduction + store)
                    const int k1 = sm.pk1[p], k2 = sm.pk2[p];
                    double R1[12], R2[12]; // poses of the pair, wave-uniform
#pragma unroll
                    for (int i = 0; i < 12; ++i) { R1[i] = uniform_f64(sm.Rt[12 * k1 + i]); R2[i] = uniform_f64(sm.Rt[12 * k2 + i]); }
                    double acc[36];
#pragma unroll
                    for (int i = 0; i < 36; ++i) acc[i] = 0;
                    if (k1 == k2) { // every edge of keyframe k1 pairs with itself: one Jacobian, symmetric 2x2 core, upper triangle only
                        // software-pipelined: streams the keyframe's own list (j) with the landmark ids two steps and the weight, the
                        // landmark, Dinv and b_l one step ahead
                        const int jbeg = sm.kfp[k1], jend = sm.kfp[k1 + 1];
                        int j = jbeg + lane;
                        double accb[6] = {0, 0, 0, 0, 0, 0}; // this keyframe's share of W Dinv b_l (reduced right-hand side)
                        if (jbeg < jend) { // (an empty list has no valid record to prefetch)
                        int ln = kf_lm[min(j, jend - 1)], lnn = kf_lm[min(j + 64, jend - 1)];
                        double wa = recW[min(j, jend - 1)];
                        double pax = PC(P, 0, ln), pay = PC(P, 1, ln), paz = PC(P, 2, ln);
                        double2 Da, Db, Dc;
                        loadD(ln, Da, Db, Dc);
                        double g0 = PC(bl, 0, ln), g1 = PC(bl, 1, ln), g2 = PC(bl, 2, ln);
                        PH(15); // (item prologue: first-row operands requested)
                        for (; j < jend; j += 64) {
                            LM_PRIO_TICK(prio_cnt);
                            double2 Dan, Dbn, Dcn;
                            loadD_row(lnn, Dan, Dbn, Dcn); // (first: see loadD_row -- nothing of THIS row's prefetch is in flight yet)
                            const int lnnn = kf_lm[min(j + 128, jend - 1)];
                            const double wan = recW[min(j + 64, jend - 1)];
                            const double paxn = PC(P, 0, lnn), payn = PC(P, 1, lnn), pazn = PC(P, 2, lnn);
                            const double g0n = PC(bl, 0, lnn), g1n = PC(bl, 1, lnn), g2n = PC(bl, 2, lnn);
                            double A1[12], B1[6];
                            double4 ra; // the linearisation record {x, y, 1/Z, w}: camera-frame part recomputed from the landmark
                            cam_norm(R1, pax, pay, paz, ra.x, ra.y, ra.z); ra.w = wa;
                            jac_norm(ra.x, ra.y, ra.z, A1);
                            jac_point_unit(ra.x, ra.y, R1, B1); // (Bt = rho B1: rho goes into l0, l1 below)
                            const double wr = ra.w * ra.z;
                            const double l0 = wr * ck.fx2, l1 = wr * ck.fy2; // = w rho fx^2, w rho fy^2
                            double BD[6];
#pragma unroll
                            for (int r = 0; r < 2; ++r) {
                                BD[3 * r] = B1[3 * r] * Da.x + B1[3 * r + 1] * Da.y + B1[3 * r + 2] * Db.x;
                                BD[3 * r + 1] = B1[3 * r] * Da.y + B1[3 * r + 1] * Db.y + B1[3 * r + 2] * Dc.x;
                                BD[3 * r + 2] = B1[3 * r] * Db.x + B1[3 * r + 1] * Dc.x + B1[3 * r + 2] * Dc.y;
                            }
                            {
                                const double m0 = l0 * (BD[0] * g0 + BD[1] * g1 + BD[2] * g2), m1 = l1 * (BD[3] * g0 + BD[4] * g1 + BD[5] * g2);
#pragma unroll
                                for (int r = 0; r < 6; ++r) accb[r] = a_fma2(A1, r, m0, m1, accb[r]);
                            }
                            const double M00 = l0 * l0 * (BD[0] * B1[0] + BD[1] * B1[1] + BD[2] * B1[2]);
                            const double M01 = l0 * l1 * (BD[0] * B1[3] + BD[1] * B1[4] + BD[2] * B1[5]);
                            const double M11 = l1 * l1 * (BD[3] * B1[3] + BD[4] * B1[4] + BD[5] * B1[5]);
#pragma unroll
                            for (int r = 0; r < 6; ++r) {
                                const double m0 = a_dot2(A1, r, M00, M01), m1 = a_dot2(A1, r, M01, M11);
#pragma unroll
                                for (int c = r; c < 6; ++c) acc[6 * r + c] = a_fma2(A1, c, m0, m1, acc[6 * r + c]);
                            }
                            wa = wan; pax = paxn; pay = payn; paz = pazn; Da = Dan; Db = Dbn; Dc = Dcn; lnn = lnnn; g0 = g0n; g1 = g1n; g2 = g2n;
                        }
                        }
                        // 21 upper-triangle sums + the 6 right-hand-side sums, one value per lane after the butterfly
                        double red[27];
                        {
                            int idx = 0;
#pragma unroll
                            for (int r = 0; r < 6; ++r)
#pragma unroll
                                for (int c = r; c < 6; ++c) red[idx++] = acc[6 * r + c];
#pragma unroll
                            for (int r = 0; r < 6; ++r) red[21 + r] = accb[r];
                        }
                        PH(22); // (rows)
                        wave_reduce_scatter<27>(red, lane);
                        if (slot27 >= 21) sm.bs[6 * k1 + slot27 - 21] = sm.bp[6 * k1 + slot27 - 21] - red[0];
                        else if (slot27 >= 0) {
                            int r = 0, rem = slot27;
                            while (rem >= 6 - r) { rem -= 6 - r; ++r; }
                            const int c = r + rem;
                            const double v = sm.Hpp[36 * k1 + 6 * r + c] + (r == c ? lambda : 0.0) - red[0];
                            sm.S[(6 * k1 + r) * np + 6 * k1 + c] = v;
                            sm.S[(6 * k1 + c) * np + 6 * k1 + r] = v;
                        }
                        continue;
                    } else {
                        const int jbeg = sm.pairp[p], jend = sm.pairp[p + 1];
                        int j = jbeg + lane;
                        if (jbeg >= jend) continue; // no common landmark: the block stays zero (S was cleared), no butterfly
                        {
                        int2 h = hits[min(j, jend - 1)];
                        int2 hn = hits[min(j + 64, jend - 1)];
                        double wa = recW[h.x & 0xFFFF], wb = recW[(unsigned)h.x >> 16];
                        double pax = PC(P, 0, h.y), pay = PC(P, 1, h.y), paz = PC(P, 2, h.y);
                        double2 Da, Db, Dc;
                        loadD(h.y, Da, Db, Dc);
                        PH(15);
                        for (; j < jend; j += 64) {
                            LM_PRIO_TICK(prio_cnt);
                            double2 Dan, Dbn, Dcn;
                            loadD_row(hn.y, Dan, Dbn, Dcn); // (first: see loadD_row)
                            const int2 hnn = hits[min(j + 128, jend - 1)];
                            const double wan = recW[hn.x & 0xFFFF], wbn = recW[(unsigned)hn.x >> 16];
                            const double paxn = PC(P, 0, hn.y), payn = PC(P, 1, hn.y), pazn = PC(P, 2, hn.y);
                            double A1[12], A2[12], B1[6], B2[6];
                            double4 ra, rb; // both observations of the landmark: 24 B of landmark + two weights instead of 64 B of records
                            cam_norm(R1, pax, pay, paz, ra.x, ra.y, ra.z); ra.w = wa;
                            cam_norm(R2, pax, pay, paz, rb.x, rb.y, rb.z); rb.w = wb;
                            jac_norm(ra.x, ra.y, ra.z, A1);
                            jac_point_unit(ra.x, ra.y, R1, B1); // (Bt = rho B: the two rhos go into ww below)
                            jac_norm(rb.x, rb.y, rb.z, A2);
                            jac_point_unit(rb.x, rb.y, R2, B2);
                            double BD[6];
#pragma unroll
                            for (int r = 0; r < 2; ++r) {
                                BD[3 * r] = B1[3 * r] * Da.x + B1[3 * r + 1] * Da.y + B1[3 * r + 2] * Db.x;
                                BD[3 * r + 1] = B1[3 * r] * Da.y + B1[3 * r + 1] * Db.y + B1[3 * r + 2] * Dc.x;
                                BD[3 * r + 2] = B1[3 * r] * Db.x + B1[3 * r + 1] * Dc.x + B1[3 * r + 2] * Dc.y;
                            }
                            const double ww = (ra.w * ra.z) * (rb.w * rb.z); // M = L1 (Bt1 D Bt2^T) L2 with L = w diag(fx^2, fy^2), Bt = rho B
                            const double lw[4] = {ww * ck.fx2 * ck.fx2, ww * ck.fx2 * ck.fy2, ww * ck.fy2 * ck.fx2, ww * ck.fy2 * ck.fy2};
                            double M[4];
#pragma unroll
                            for (int r = 0; r < 2; ++r)
#pragma unroll
                                for (int c = 0; c < 2; ++c) M[2 * r + c] = lw[2 * r + c] * (BD[3 * r] * B2[3 * c] + BD[3 * r + 1] * B2[3 * c + 1] + BD[3 * r + 2] * B2[3 * c + 2]);
#pragma unroll
                            for (int r = 0; r < 6; ++r) {
                                const double m0 = a_dot2(A1, r, M[0], M[2]), m1 = a_dot2(A1, r, M[1], M[3]);
#pragma unroll
                                for (int c = 0; c < 6; ++c) acc[6 * r + c] = a_fma2(A2, c, m0, m1, acc[6 * r + c]);
                            }
                            wa = wan; wb = wbn; pax = paxn; pay = payn; paz = pazn; Da = Dan; Db = Dbn; Dc = Dcn; hn = hnn;
                        }
                        }
                    }
                    PH(22);
                    wave_reduce_scatter<36>(acc, lane);
                    if (slot36 >= 0) {
                        const int r = slot36 / 6, c = slot36 - 6 * r;
                        sm.S[(6 * k1 + r) * np + 6 * k2 + c] = -acc[0];
                        sm.S[(6 * k2 + c) * np + 6 * k1 + r] = -acc[0];
                    }
                }
                __syncthreads();
                PH(6);
                // Cholesky S = L L^T, RIGHT-looking over 6x6 block columns (np = 6 nk), two barriers per block column:
                //   P1  every lane that owns a row of the panel below block (J, J) factors the 6x6 diagonal block in registers (redundantly:
                //       cheaper than one lane + a broadcast) and solves its row against it, in place; the right-hand side rides along as one
                //       more row (L y = b is solved by the factorisation itself: only the backward substitution is left afterwards)
                //   P2  the trailing blocks (I, K), J < K <= I, take  -= X_I X_K^T  with one lane per block ROW: 6 outputs from 6 + 36 operands.
                // The left-looking form it replaces recomputed every element of a block column as a dot product over all earlier columns with one
                // lane per ELEMENT: 2 LDS operands per FMA, and its column update was bound by LDS bandwidth (in-kernel clocks, per block column:
                // update 1.84 k cycles, diagonal + rows 2.4 k, barriers 0.55 k).  L_JJ goes to a side array (Ld): nothing reads it before the solve.
                for (int J = 0; J < nk && ok2; ++J) {
                    const int m = nk - J - 1, nrows = m * 6;
                    const bool rhs_row = tid == kLmBlock - 1;
                    if (tid < nrows || tid == 0 || rhs_row) {
                        double D[21]; // lower triangle, row-major: (i,j) -> i*(i+1)/2 + j
#pragma unroll
                        for (int i = 0; i < 6; ++i)
#pragma unroll
                            for (int j = 0; j <= i; ++j) D[i * (i + 1) / 2 + j] = sm.S[(6 * J + i) * np + 6 * J + j];
                        bool good = true;
                        double rd[6];
#pragma unroll
                        for (int j = 0; j < 6; ++j) {
                            double d = D[j * (j + 1) / 2 + j];
#pragma unroll
                            for (int kk = 0; kk < j; ++kk) d -= D[j * (j + 1) / 2 + kk] * D[j * (j + 1) / 2 + kk];
                            if (!(d > 0.0) || !isfinite(d)) good = false;
                            rd[j] = rsqrt_nr(d); // the dependent chain of the factorisation: no IEEE sqrt / division on it
                            D[j * (j + 1) / 2 + j] = d * rd[j];
#pragma unroll
                            for (int i = j + 1; i < 6; ++i) {
                                double v = D[i * (i + 1) / 2 + j];
#pragma unroll
                                for (int kk = 0; kk < j; ++kk) v -= D[i * (i + 1) / 2 + kk] * D[j * (j + 1) / 2 + kk];
                                D[i * (i + 1) / 2 + j] = v * rd[j];
                            }
                        }
                        if (tid < nrows || rhs_row) {
                            double* rowv = rhs_row ? &sm.bs[6 * J] : &sm.S[(6 * (J + 1) + tid) * np + 6 * J];
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
                    PH(16);
                    __syncthreads();
                    PH(17);
                    if (sm.flag[1]) { ok2 = false; break; } // uniform
                    // P2: trailing update.  Items: (block pair (a, b), a >= b, of the m block rows below; row r) and, for the right-hand side, one
                    // item per block b
                    const int npair = m * (m + 1) / 2, nitem = npair * 6 + m;
                    for (int t = tid; t < nitem; t += kLmBlock) {
                        const double* xi; double* out; int Kb;
                        if (t < npair * 6) {
                            const int pr = t / 6, r = t - 6 * pr;
                            int a = 0, rem = pr;
                            while (rem > a) { rem -= a + 1; ++a; }
                            const int Ib = J + 1 + a; Kb = J + 1 + rem;
                            xi = &sm.S[(6 * Ib + r) * np + 6 * J]; out = &sm.S[(6 * Ib + r) * np + 6 * Kb];
                        } else { Kb = J + 1 + (t - npair * 6); xi = &sm.bs[6 * J]; out = &sm.bs[6 * Kb]; }
                        const double2* xi2 = reinterpret_cast<const double2*>(xi);
                        const double2 a0 = xi2[0], a1 = xi2[1], a2 = xi2[2];
                        double2* o2 = reinterpret_cast<double2*>(out);
                        double2 o[3] = {o2[0], o2[1], o2[2]};
                        double v[6] = {o[0].x, o[0].y, o[1].x, o[1].y, o[2].x, o[2].y};
#pragma unroll
                        for (int c = 0; c < 6; ++c) {
                            const double2* xk = reinterpret_cast<const double2*>(&sm.S[(6 * Kb + c) * np + 6 * J]);
                            const double2 b0 = xk[0], b1 = xk[1], b2 = xk[2];
                            v[c] -= (a0.x * b0.x + a0.y * b0.y) + (a1.x * b1.x + a1.y * b1.y) + (a2.x * b2.x + a2.y * b2.y);
                        }
                        o2[0] = make_double2(v[0], v[1]); o2[1] = make_double2(v[2], v[3]); o2[2] = make_double2(v[4], v[5]);
                    }
                    PH(18);
                    __syncthreads();
                    PH(19);
                }
                PH(7);
                if (sm.flag[1]) ok2 = false;
                __syncthreads();
                PH(20);
                if (tid == 0) sm.flag[1] = 0;
                if (ok2) {
                    // backward substitution L^T x = y by wave 0, block by block from the last: lane i carries t_i = y_i - sum over the finished
                    // unknowns (and t_{i+64}: np <= 128).  Per block: its six right-hand sides go to every lane (readlane), every lane solves the
                    // 6x6 triangle L_JJ^T x = t redundantly (L_JJ from the side array, wave-uniform reads), and the unknowns below take
                    // t_i -= sum_c L[6J+c][i] x_c with the six rows of L read along i (consecutive lanes, consecutive addresses).
                    if (wave == 0) {
                        double u0 = lane < np ? sm.bs[lane] : 0.0, u1 = lane + 64 < np ? sm.bs[lane + 64] : 0.0;
                        for (int J = nk - 1; J >= 0; --J) {
                            double Lr0[6], Lr1[6], Ld[21], rdj[6], t[6], x[6];
#pragma unroll
                            for (int c = 0; c < 6; ++c) {
                                const int r = 6 * J + c;
                                Lr0[c] = lane < 6 * J ? sm.S[r * np + lane] : 0.0;
                                Lr1[c] = (6 * J > 64 && lane + 64 < 6 * J) ? sm.S[r * np + lane + 64] : 0.0;
                                rdj[c] = sm.rdiag[r];
                            }
#pragma unroll
                            for (int i = 0; i < 21; ++i) Ld[i] = sm.Ld[24 * J + i];
#pragma unroll
                            for (int c = 0; c < 6; ++c) {
                                const int r = 6 * J + c; // wave-uniform
                                t[c] = r >= 64 ? readlane_f64(u1, r - 64) : readlane_f64(u0, r);
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
                    PH(21);
                } else {
                    for (int i = tid; i < np; i += kLmBlock) sm.xp[i] = 0;
                }
                __syncthreads();
            } else {
                // pose-only: block-diagonal system, one thread per pose
                if (tid < nk) {
                    double x[6];
                    const bool ok = chol6_solve(&sm.Hpp[36 * tid], lambda, &sm.bp[6 * tid], x);
                    if (!ok) sm.flag[1] = 1;
#pragma unroll
                    for (int r = 0; r < 6; ++r) sm.xp[6 * tid + r] = x[r];
                }
                __syncthreads();
                if (sm.flag[1]) { ok2 = false; }
                __syncthreads();
                if (!ok2) { for (int i = tid; i < np; i += kLmBlock) sm.xp[i] = 0; if (tid == 0) sm.flag[1] = 0; }
                __syncthreads();
            }
            PH(8);
            // ---- update: landmarks (back-substitution) and poses; computeScale
            double scale_part = 0;
            if (with_lm) {
                for (int l0 = tid; l0 < nl; l0 += kLmU * kLmBlock) {
                    int cn[kLmU];
                    double px[kLmU], py[kLmU], pz[kLmU], g0[kLmU], g1[kLmU], g2[kLmU], Dq[kLmU][6];
                    bool on[kLmU], in[kLmU];
                    int kk[kLmU][kLmE]; float2 zz[kLmU][kLmE];
#pragma unroll
                    for (int u = 0; u < kLmU; ++u) {
                        const int l = min(l0 + u * kLmBlock, nl - 1);
                        in[u] = l0 + u * kLmBlock < nl;
                        double2 da, dbb, dc;
                        loadD_row(l, da, dbb, dc); // (first: behind the batch's global loads its LDS reads would wait for all of them)
                        Dq[u][0] = da.x; Dq[u][1] = da.y; Dq[u][2] = dbb.x; Dq[u][3] = dbb.y; Dq[u][4] = dc.x; Dq[u][5] = dc.y;
                        cn[u] = lcnt[l];
                        const int rm = slot_width(sm.rowmax, l0 - lane + u * kLmBlock); // (uniform)
#pragma unroll
                        for (int q = 0; q < kLmE; ++q) { const size_t qo = (size_t)(q < rm ? q : 0) * nl + l; kk[u][q] = skf[qo]; zz[u][q] = suv[qo]; }
                        px[u] = PC(P, 0, l); py[u] = PC(P, 1, l); pz[u] = PC(P, 2, l);
                        g0[u] = PC(bl, 0, l); g1[u] = PC(bl, 1, l); g2[u] = PC(bl, 2, l);
                    }
#pragma unroll
                    for (int u = 0; u < kLmU; ++u) on[u] = cn[u] > 0 && in[u];
#pragma unroll
                    for (int u = 0; u < kLmU; ++u) {
                        if (!in[u]) continue;
                        const int l = l0 + u * kLmBlock;
                        if (!on[u]) { PC(Pt, 0, l) = px[u]; PC(Pt, 1, l) = py[u]; PC(Pt, 2, l) = pz[u]; continue; }
                        double c0 = g0[u], c1 = g1[u], c2 = g2[u];
                        auto sub_edge = [&](int k, float2 z) {
                            double x, y, ri, wg, ex, ey, c, rho, A[12], B[6];
                            cam_norm(&sm.Rt[12 * k], px[u], py[u], pz[u], x, y, ri);
                            eval_obs(ck, x, y, z, delta, ex, ey, c, rho, wg);
                            jac_norm(x, y, ri, A);
                            jac_point_norm(x, y, ri, &sm.Rt[12 * k], B);
                            // W^T xp = w B^T (A xp_k)
                            double a0 = 0, a1 = 0;
#pragma unroll
                            for (int r = 0; r < 6; ++r) { // A[1] = A[6] = 0
                                const double xr = sm.xp[6 * k + r];
                                if (r != 1) a0 = fma(A[r], xr, a0);
                                if (r != 0) a1 = fma(A[6 + r], xr, a1);
                            }
                            a0 *= wg * ck.fx2; a1 *= wg * ck.fy2;
                            c0 -= B[0] * a0 + B[3] * a1; c1 -= B[1] * a0 + B[4] * a1; c2 -= B[2] * a0 + B[5] * a1;
                        };
#pragma unroll
                        for (int q = 0; q < kLmE; ++q) if (q < cn[u]) sub_edge(kk[u][q], zz[u][q]);
                        if (cn[u] > kLmE) { // (as in the landmark pass: the tail in one batch)
                            constexpr int kTail = kMaxKf - kLmE;
                            const int b0 = lm_ptr[l];
                            int kt[kTail]; float2 zt[kTail];
#pragma unroll
                            for (int q = 0; q < kTail; ++q) { const int e = min(b0 + kLmE + q, max(ne - 1, 0)); kt[q] = kfi[e]; zt[q] = uv2[e]; }
#pragma unroll
                            for (int q = 0; q < kTail; ++q) if (kLmE + q < cn[u]) sub_edge(kt[q], zt[q]);
                            for (int e = b0 + kMaxKf; e < b0 + cn[u]; ++e) sub_edge(kfi[e], uv2[e]);
                        }
                        const double x0 = Dq[u][0] * c0 + Dq[u][1] * c1 + Dq[u][2] * c2;
                        const double x1 = Dq[u][1] * c0 + Dq[u][3] * c1 + Dq[u][4] * c2;
                        const double x2 = Dq[u][2] * c0 + Dq[u][4] * c1 + Dq[u][5] * c2;
                        PC(Pt, 0, l) = px[u] + x0; PC(Pt, 1, l) = py[u] + x1; PC(Pt, 2, l) = pz[u] + x2;
                        scale_part += x0 * (lambda * x0 + g0[u]) + x1 * (lambda * x1 + g1[u]) + x2 * (lambda * x2 + g2[u]);
                    }
                }
            }
            if (tid < np) scale_part += sm.xp[tid] * (lambda * sm.xp[tid] + sm.bp[tid]);
            if (tid < nk) {
                double E[7];
                se3::exp(&sm.xp[6 * tid], E);
                se3::mul(E, &sm.T[7 * tid], &sm.TTrial[7 * tid]);
                expand_pose(&sm.TTrial[7 * tid], &sm.RtTrial[12 * tid]);
            }
            scale = block_sum(scale_part, sm.red) + 1e-3;
            } // !boot
            PH(9);
            double tempChi = eval(boot ? sm.Rt : sm.RtTrial, (boot || !with_lm) ? P : Pt, boot ? recW : recW_alt, boot ? sm.Hpp : sm.HppT, boot ? sm.bp : sm.bpT);
            if (boot) { currentChi = tempChi; break; }
            last_trial_is_current = false;
            PH(10);
            if (!ok2) tempChi = 1.7976931348623157e308;
            rho_gain = (currentChi - tempChi) / scale;
            const bool accept = rho_gain > 0 && isfinite(tempChi); // uniform: all inputs are block-uniform
            if (accept) {
                double alpha = 1. - pow(2 * rho_gain - 1, 3);
                alpha = fmin(alpha, 2. / 3.);
                lambda *= fmax(1. / 3., alpha);
                ni = 2;
                currentChi = tempChi;
                __syncthreads();
                for (int i = tid; i < nk * 7; i += kLmBlock) sm.T[i] = sm.TTrial[i];
                for (int i = tid; i < nk * 12; i += kLmBlock) sm.Rt[i] = sm.RtTrial[i];
                if (with_lm) { double* t = P; P = Pt; Pt = t; }
                { double* ta = recW; recW = recW_alt; recW_alt = ta; last_trial_is_current = true; }
                for (int i = tid; i < nk * 36; i += kLmBlock) sm.Hpp[i] = sm.HppT[i];
                for (int i = tid; i < np; i += kLmBlock) sm.bp[i] = sm.bpT[i];
                __syncthreads();
            } else {
                lambda *= ni;
                ni *= 2;
            }
            ++qmax;
            again = (rho_gain < 0) && qmax < 10;
        }
        } // trials
        if (boot) continue;
        total_trials += qmax;
        if (st && tid == 0 && it < VSLAM_LM_MAX_ITERS) { st->chi2_iter[it] = currentChi; st->lambda_iter[it] = lambda; st->trials_iter[it] = qmax; }
        if (qmax == 10 || rho_gain == 0) { ++it; bound = 0; break; }
    }
    if (st && tid == 0) { st->iterations = it; st->total_trials = total_trials; st->chi2_final = currentChi; st->lambda_final = lambda; }

    PH(11);
    if (last_trial_is_current) chi_pass(sm.Rt, P); else chi_pass(sm.RtTrial, with_lm ? Pt : P);
    // ------------------------------------------------------------------ chi2 classification (optimization.cpp:224-266)
    int newly_flagged = 0;   // this thread cleared the flag of a landmark that entered the pass flagged in
    if (classify && !IMPL) {
        double th = delta; // optimization.cpp:154: chi2_th is both the Huber delta and the initial classification threshold
        for (int iteration = 0; iteration < 5; ++iteration) {
            double out = 0, in = 0;
            for (int j = tid; j < ntot; j += kLmBlock) {
                if (chi2k[j] > th) out += 1; else in += 1;
            }
            out = block_sum(out, sm.red);
            in = block_sum(in, sm.red);
            const double ratio = in / (in + out);
            if (ratio > 0.5) break;
            th *= 2;
        }
        // last edge of the landmark wins (ascending edge order).  lm_ptr -> kf_pos -> chi2 is a chain of three dependent loads: four
        // landmarks per lane per trip, level by level (a strided loop is left serial: one round trip per load per landmark)
        for (int l0 = tid; l0 < nl; l0 += 4 * kLmBlock) {
            int la[4], pe[4], pp[4];
            uint8_t av[4];
            double cv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { la[u] = min(l0 + u * kLmBlock, nl - 1); av[u] = act[la[u]]; pe[u] = lm_ptr[la[u] + 1]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) pp[u] = kf_pos[min(max(pe[u] - 1, 0), max(ne - 1, 0))];
#pragma unroll
            for (int u = 0; u < 4; ++u) cv[u] = chi2k[min(max(pp[u], 0), max(ne - 1, 0))];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (l0 + u * kLmBlock < nl && av[u]) { const bool keep = !(cv[u] > th); a.lm_inlier[lm0 + la[u]] = keep; newly_flagged |= !keep; } // (active = flagged in on entry)
        }
        if (tid == 0 && a.chi2_thr) a.chi2_thr[w] = th;
    }
    if (SCHED && pass < 2) {
        if (!done) { // (done: this was the continuation -- the last pass, whatever its own classification flagged)
            const bool repeatable = __syncthreads_or(newly_flagged) == 0; // (uniform) the next pass would see the inputs this one saw
            if (repeatable) {
                done = true; update_poses = 1;
                if (bound > 0 && bound < kSchedFinalIters) { bound = kSchedFinalIters; continue; } // continue it as the last pass
                // (bound == 0: the loop stopped by its own rule; the last pass would stop there too -- this state is the result)
            }
        }
    }
    break;
    }
    // ------------------------------------------------------------------ write-back (:272-287, :429-435)
    __syncthreads();
    if (update_poses) for (int i = tid; i < nk * 7; i += kLmBlock) a.T[Tbase + i] = sm.T[i];
    if (!IMPL && ka.want_chi2) // chi2 back to the caller's edge order (edges of excluded landmarks: 0)
        for (int e = tid; e < ne; e += kLmBlock) chi2[e] = act[lmi[e]] ? chi2k[kf_pos[e]] : 0.0;
    if (with_lm && update_lms)
        for (int l = tid; l < nl; l += kLmBlock)
            if (act[l]) { a.xyz[3 * ((size_t)lm0 + l)] = (float)PC(P, 0, l); a.xyz[3 * ((size_t)lm0 + l) + 1] = (float)PC(P, 1, l); a.xyz[3 * ((size_t)lm0 + l) + 2] = (float)PC(P, 2, l); }
    PH(12);
    } // passes
    if (SCHED && threadIdx.x == 0) ka.passes[w] = pass; // passes executed: 1 or 2 = an early pass was continued as the last one
    if (threadIdx.x == 0) ka.status[w] = VSLAM_OK;
#undef PH
#undef PC
}

// reprojection-error inlier test of the motion-only stage (solvePnPRansac's reprojectionError contract)
__global__ __launch_bounds__(256) void pnp_inlier_kernel(const float* __restrict__ xyz, const float* __restrict__ uv, const int32_t* __restrict__ d_n,
                                                        int capacity, const double* __restrict__ d_T, const double K0, const double K1,
                                                        const double K2, const double K3, double thr2, uint8_t* __restrict__ inlier,
                                                        int32_t* __restrict__ n_inl) {
    const int b = blockIdx.x;
    const int n = min(max(d_n[b], 0), capacity);
    __shared__ double Rt[12];
    __shared__ int cnt;
    if (threadIdx.x == 0) { expand_pose(d_T + 7 * b, Rt); cnt = 0; }
    __syncthreads();
    const double K[4] = {K0, K1, K2, K3};
    int mine = 0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const size_t g = (size_t)b * capacity + i;
        double X, Y, Z, ex, ey;
        project_err(Rt, K, (double)xyz[3 * g], (double)xyz[3 * g + 1], (double)xyz[3 * g + 2], uv[2 * g], uv[2 * g + 1], X, Y, Z, ex, ey);
        const double c = ex * ex + ey * ey;
        const bool ok = isfinite(c) && c <= thr2;
        if (inlier) inlier[g] = ok;
        mine += ok;
    }
    atomicAdd(&cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0 && n_inl) n_inl[b] = cnt;
}

// ------------------------------------------------------------------------------------------- lean motion-only LM
// pnp_wave_kernel: ONE WAVE per single-pose problem (the motion-only stage of VO::motion_estimation, and the refinement of the RANSAC
// pose on its inliers).  The window kernel above spends a 512-thread workgroup, its list builders and ~25 barriers per iteration on a
// problem with one 6x6 block and a few hundred points (~40 k cycles per iteration, almost all of it barriers and serial code); here the
// points are dealt to the 64 lanes, the 27 sums of the normal equations come out of one halving butterfly, every lane then holds all of
// them (v_readlane from the lane the butterfly left each sum in) and factors the 6x6 system redundantly -- no LDS, no barrier.  Same
// algorithm as the window kernel in mode 1 with one keyframe (g2o Levenberg: lambda0 = 1e-5 max diag H, <= 10 trials per iteration,
// rho = (chi2 - chi2_trial) / (dx (lambda dx + b) + 1e-3), accept: lambda *= max(1/3, min(2/3, 1 - (2 rho - 1)^3)), reject:
// lambda *= ni, ni *= 2), same residual / Jacobian helpers, fixed summation order; followed by the reprojection-error inlier test.
__global__ __launch_bounds__(64) void pnp_wave_kernel(const float* __restrict__ xyz, const float* __restrict__ uv, const int32_t* __restrict__ d_n, int capacity,
                                                     double* __restrict__ d_T, int iters, double K0, double K1, double K2, double K3, double delta,
                                                     double thr2, uint8_t* __restrict__ inlier, int32_t* __restrict__ n_inl, vslam_lm_stats* __restrict__ stats) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int n = min(max(d_n[b], 0), capacity);
    const float* px = xyz + 3 * (size_t)b * capacity;
    const float2* pz = reinterpret_cast<const float2*>(uv) + (size_t)b * capacity;
    const double K[4] = {K0, K1, K2, K3};
    const CamK ck = make_camk(K);
    vslam_lm_stats* st = stats ? stats + b : nullptr;
    double T[7], Rt[12];
#pragma unroll
    for (int i = 0; i < 7; ++i) T[i] = d_T[7 * (size_t)b + i];
    expand_pose(T, Rt);
    const int slot27 = wave_slot<27>(lane);
    // robust cost at pose R; with LIN also H (upper triangle, 21) and b (6) of the normal equations, left in every lane
    auto evaluate = [&](const double* R, bool lin, double (&H)[36], double (&g)[6]) -> double {
        double part = 0, acc[27];
#pragma unroll
        for (int i = 0; i < 27; ++i) acc[i] = 0;
        for (int i0 = 0; i0 < n; i0 += 128) { // two points per lane in flight
            const int ia = i0 + lane, ib = i0 + 64 + lane;
            const int ja = min(ia, n - 1), jb = min(ib, n - 1);
            const double ax = px[3 * ja], ay = px[3 * ja + 1], az = px[3 * ja + 2], bx = px[3 * jb], by = px[3 * jb + 1], bz = px[3 * jb + 2];
            const float2 za = pz[ja], zb = pz[jb];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if ((u ? ib : ia) >= n) continue;
                double x, y, ri, wgt, ex, ey, c, rho;
                cam_norm(R, u ? bx : ax, u ? by : ay, u ? bz : az, x, y, ri);
                eval_obs(ck, x, y, u ? zb : za, delta, ex, ey, c, rho, wgt);
                part += rho;
                if (lin) {
                    double A[12], wA[12];
                    jac_norm(x, y, ri, A);
                    const double l0 = wgt * ck.fx2, l1 = wgt * ck.fy2;
#pragma unroll
                    for (int i = 0; i < 6; ++i) { wA[i] = l0 * A[i]; wA[6 + i] = l1 * A[6 + i]; }
                    int idx = 0;
#pragma unroll
                    for (int r = 0; r < 6; ++r)
#pragma unroll
                        for (int cc = r; cc < 6; ++cc) { acc[idx] = a_fma_pair(wA, r, A, cc, acc[idx]); ++idx; }
#pragma unroll
                    for (int r = 0; r < 6; ++r) acc[21 + r] = a_fma2(wA, r, -ex, -ey, acc[21 + r]);
                }
            }
        }
        const double total = wave_sum(part);
        if (lin) {
            wave_reduce_scatter<27>(acc, lane); // acc[0] = sum number slot27 (lane)
            double v[27];
#pragma unroll
            for (int sidx = 0; sidx < 27; ++sidx) {
                const int src = __ffsll((long long)__ballot(slot27 == sidx)) - 1; // wave-uniform
                v[sidx] = readlane_f64(acc[0], src);
            }
            int idx = 0;
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int cc = r; cc < 6; ++cc) { H[6 * r + cc] = v[idx]; H[6 * cc + r] = v[idx]; ++idx; }
#pragma unroll
            for (int r = 0; r < 6; ++r) g[r] = v[21 + r];
        }
        return total;
    };
    double H[36], g[6], lambda = 0, ni = 2, currentChi = 0;
    int it = 0, total_trials = 0;
    for (it = 0; it < iters; ++it) {
        currentChi = evaluate(Rt, true, H, g);
        if (it == 0) {
            if (st && lane == 0) st->chi2_init = currentChi;
            double md = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a) md = fmax(fabs(H[7 * a]), md);
            lambda = 1e-5 * md;
            ni = 2;
        }
        double rho_gain = 0;
        int qmax = 0;
        bool again = true;
        while (again) {
            double x[6], E[7], Tt[7], Rtt[12], Hd[36], gd[6];
            const bool ok2 = chol6_solve(H, lambda, g, x);
            if (!ok2) {
#pragma unroll
                for (int i = 0; i < 6; ++i) x[i] = 0;
            }
            se3::exp(x, E);
            se3::mul(E, T, Tt);
            expand_pose(Tt, Rtt);
            double tempChi = evaluate(Rtt, false, Hd, gd);
            if (!ok2) tempChi = 1.7976931348623157e308;
            double scale = 1e-3;
#pragma unroll
            for (int i = 0; i < 6; ++i) scale += x[i] * (lambda * x[i] + g[i]);
            rho_gain = (currentChi - tempChi) / scale;
            if (rho_gain > 0 && isfinite(tempChi)) {
                double alpha = 1. - pow(2 * rho_gain - 1, 3);
                alpha = fmin(alpha, 2. / 3.);
                lambda *= fmax(1. / 3., alpha);
                ni = 2;
                currentChi = tempChi;
#pragma unroll
                for (int i = 0; i < 7; ++i) T[i] = Tt[i];
#pragma unroll
                for (int i = 0; i < 12; ++i) Rt[i] = Rtt[i];
            } else {
                lambda *= ni;
                ni *= 2;
            }
            ++qmax;
            again = (rho_gain < 0) && qmax < 10;
        }
        total_trials += qmax;
        if (st && lane == 0 && it < VSLAM_LM_MAX_ITERS) { st->chi2_iter[it] = currentChi; st->lambda_iter[it] = lambda; st->trials_iter[it] = qmax; }
        if (qmax == 10 || rho_gain == 0) { ++it; break; }
    }
    if (st && lane == 0) { st->iterations = it; st->total_trials = total_trials; st->chi2_final = currentChi; st->lambda_final = lambda; }
    if (lane < 7) d_T[7 * (size_t)b + lane] = T[lane];
    // inliers at the estimate: reprojection error <= reproj_thr (the contract of pnp_inlier_kernel)
    int mine = 0;
    for (int i = lane; i < n; i += 64) {
        const size_t gi = (size_t)b * capacity + i;
        double X, Y, Z, ex, ey;
        project_err(Rt, K, (double)px[3 * i], (double)px[3 * i + 1], (double)px[3 * i + 2], uv[2 * gi], uv[2 * gi + 1], X, Y, Z, ex, ey);
        const double c = ex * ex + ey * ey;
        const bool ok = isfinite(c) && c <= thr2;
        if (inlier) inlier[gi] = ok;
        mine += ok;
    }
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
    if (lane == 0 && n_inl) n_inl[b] = mine;
}

// ------------------------------------------------------------------------------------------- lean pose-only pass of the schedule
// pose_only_wave_kernel: optimize_pose_only (optimization.cpp:290-436) as the LAST pass of the per-keyframe schedule (run_vslam.cpp:67-70),
// one workgroup per window, ONE WAVE PER KEYFRAME.  With the landmarks fixed the keyframes only share lambda, the gain ratio and the
// accept / reject decision; the window kernel nevertheless ran its general machinery for it (keyframe-major lists with the Schur hit lists'
// bookkeeping, 8 waves dealt over rows of all keyframes, per-(keyframe, wave) partial sums, ~10 barriers per iteration: 0.57 ms of the
// 4.15 ms schedule).  Here a setup of two ballot scans cuts the window's active edges (landmark still an inlier, :334) into per-keyframe
// runs -- edge id, observation and the landmark position (f32 at rest, constant in this pass) copied keyframe-major, so the iteration
// streams 24 contiguous bytes per edge and gathers nothing -- and wave k then runs keyframe k's motion-only problem exactly like
// pnp_wave_kernel: 27 sums through one butterfly straight into LDS, 6x6 solve in every lane, trial evaluation that also linearises (an
// accepted trial needs no second pass).  Per trial two barriers: the "some 6x6 factorisation failed" flag, and the per-keyframe chi2 /
// scale partials, summed in keyframe order by every wave.  Then the chi2 classification (:224-266) -- at the state of the last EVALUATED
// trial, g2o's quirk, like the window kernel.  Runs only behind the three optimize_map passes of a schedule: they have validated the graph
// (status word) and nothing here needs the landmark CSR.
constexpr int kPoBlock = 64 * kMaxKf;
struct PoShared {
    double H[2][kMaxKf][36], g[2][kMaxKf][6];
    double part[2][kMaxKf], spart[2][kMaxKf];
    int cnt[kMaxKf][kMaxKf];
    int kbeg[kMaxKf + 1];
    int flag[2];
    int cin[2][kMaxKf], cout[2][kMaxKf];
    double T[2][kMaxKf][7], R[3][kMaxKf][12];
};
__global__ __launch_bounds__(kPoBlock) void pose_only_wave_kernel(LmKernelArgs ka, int iters, int update_poses) {
    const LmWindowArgs& a = ka.a;
    __shared__ PoShared sm;
    const int w = ka.order ? ka.order[blockIdx.x] : (int)blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (ka.status[w] != VSLAM_OK) return; // uniform: the first pass of the schedule rejected this window
    long long* cyc = ka.dbg_cycles ? ka.dbg_cycles + kDbgSlots * (size_t)w : nullptr; // tuning aid (VSLAM_LM_PROFILE=1): slots 0..8 of this pass
    long long t_ph = cyc ? clock64() : 0;
#define POH(i) do { if (cyc && tid == 0) { const long long t1__ = clock64(); cyc[i] += t1__ - t_ph; t_ph = t1__; } } while (0)
    const int nk = a.n_kf_w ? min(max(a.n_kf_w[w], 1), a.n_kf) : a.n_kf; // (a.n_kf: the pose stride)
    const int lm0 = a.lm_off[w], e0 = a.edge_off[w], ne = a.edge_off[w + 1] - e0;
    const int32_t* kfi = a.kf_idx + e0;
    const int32_t* lmi = a.lm_idx + e0;
    const float2* uv2 = reinterpret_cast<const float2*>(a.uv) + e0;
    const float* xyz = a.xyz + 3 * (size_t)lm0;
    uint8_t* inl = a.lm_inlier + lm0;
    int32_t* list = a.kf_edges + e0;                                   // keyframe-major position -> edge id
    float2* uvk = reinterpret_cast<float2*>(ka.uvk) + e0;              // ... -> observation
    float4* posk = reinterpret_cast<float4*>(a.lin) + e0;              // ... -> landmark position (16 B of the 16 B-per-edge linearisation scratch)
    double* chik = ka.chi2k + e0;                                      // ... -> chi2 at the final state
    const double K[4] = {a.K[0], a.K[1], a.K[2], a.K[3]};
    const CamK ck = make_camk(K);
    const double delta = a.huber_delta;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    // ---- setup: per-keyframe runs of the active edges (ascending edge id inside a keyframe), two ballot scans over this wave's chunk
    const int per = (((ne + kMaxKf - 1) / kMaxKf) + 63) & ~63;
    const int c0 = min(wave * per, ne), c1 = min(c0 + per, ne);
    {
        int cnt[kMaxKf];
#pragma unroll
        for (int k = 0; k < kMaxKf; ++k) cnt[k] = 0;
        for (int base = c0; base < c1; base += 256) { // four rows per trip: ids first, then the flags they point at (two dependent round trips per TRIP)
            int kq[4], lq[4]; bool vq[4]; uint8_t iq[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int e = base + 64 * u + lane; vq[u] = e < c1; const int es = min(e, ne - 1); kq[u] = kfi[es]; lq[u] = lmi[es]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) iq[u] = inl[lq[u]];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool act = vq[u] && iq[u] != 0;
#pragma unroll
                for (int kk = 0; kk < kMaxKf; ++kk) cnt[kk] += __popcll(__ballot(act && kq[u] == kk));
            }
        }
        if (lane < kMaxKf) {
            int v = 0;
#pragma unroll
            for (int kk = 0; kk < kMaxKf; ++kk) if (lane == kk) v = cnt[kk];
            sm.cnt[wave][lane] = v;
        }
        if (tid < 2) sm.flag[tid] = 0;
    }
    __syncthreads();
    int off[kMaxKf]; // write offset of this wave's chunk inside every keyframe's run
    {
        int run = 0;
#pragma unroll
        for (int k = 0; k < kMaxKf; ++k) {
            int before = 0, tot = 0;
#pragma unroll
            for (int c = 0; c < kMaxKf; ++c) { const int v = sm.cnt[c][k]; tot += v; if (c < wave) before += v; }
            off[k] = run + before;
            if (tid == 0) sm.kbeg[k] = run;
            run += tot;
        }
        if (tid == 0) sm.kbeg[kMaxKf] = run;
    }
    for (int base = c0; base < c1; base += 256) {
        int kq[4], lq[4], eq[4]; bool vq[4]; uint8_t iq[4]; float2 zq[4]; float Xq[4], Yq[4], Zq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = base + 64 * u + lane; vq[u] = e < c1; eq[u] = e;
            const int es = min(e, ne - 1);
            kq[u] = kfi[es]; lq[u] = lmi[es]; zq[u] = uv2[es];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { iq[u] = inl[lq[u]]; Xq[u] = xyz[3 * lq[u]]; Yq[u] = xyz[3 * lq[u] + 1]; Zq[u] = xyz[3 * lq[u] + 2]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool act = vq[u] && iq[u] != 0;
#pragma unroll
            for (int kk = 0; kk < kMaxKf; ++kk) {
                const unsigned long long m = __ballot(act && kq[u] == kk);
                if (act && kq[u] == kk) {
                    const int pos = off[kk] + __popcll(m & lt_mask);
                    list[pos] = eq[u]; uvk[pos] = zq[u]; posk[pos] = make_float4(Xq[u], Yq[u], Zq[u], 0.f);
                }
                off[kk] += __popcll(m);
            }
        }
    }
    __syncthreads();
    POH(0);
    const int ntot = sm.kbeg[kMaxKf];
    const bool mine = wave < nk;                          // this wave owns keyframe `wave`
    const int kb = mine ? sm.kbeg[wave] : 0, ke = mine ? sm.kbeg[wave + 1] : 0;
    // poses live in LDS (current / trial / last evaluated trial per keyframe) and reach the evaluation loop through scalar registers:
    // held in VGPRs next to the 27 accumulators they spilled (168 registers per lane at 12 waves per workgroup)
    {
        double T[7], Rt[12];
#pragma unroll
        for (int i = 0; i < 7; ++i) T[i] = a.T[((size_t)w * a.n_kf + min(wave, nk - 1)) * 7 + i];
        expand_pose(T, Rt);
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 7; ++i) sm.T[0][wave][i] = T[i];
#pragma unroll
            for (int i = 0; i < 12; ++i) { sm.R[0][wave][i] = Rt[i]; sm.R[2][wave][i] = Rt[i]; }
        }
        __builtin_amdgcn_wave_barrier();
    }
    const int slot27 = wave_slot<27>(lane);
    // robust cost of this keyframe's edges at pose R; LIN: its 6x6 block and right-hand side into sm.H[buf][wave], sm.g[buf][wave];
    // CHI: chi2 per edge into chik (keyframe-major)
    auto evaluate = [&](const double* Rlds, bool lin, int buf, bool store_chi) -> double {
        double R[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) R[i] = uniform_f64(Rlds[i]);
        double part = 0, acc[27];
#pragma unroll
        for (int i = 0; i < 27; ++i) acc[i] = 0;
        // two edges per lane per trip; the operands of the NEXT trip are requested before this trip's arithmetic (three waves per SIMD
        // do not cover a memory round trip per trip on their own)
        float4 pp[2], pn[2]; float2 zz[2], zn[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) { const int j = min(kb + 64 * u + lane, max(ke - 1, 0)); pp[u] = posk[j]; zz[u] = uvk[j]; }
        for (int j0 = kb; j0 < ke; j0 += 128) {
#pragma unroll
            for (int u = 0; u < 2; ++u) { const int j = min(j0 + 128 + 64 * u + lane, ke - 1); pn[u] = posk[j]; zn[u] = uvk[j]; }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int j = j0 + 64 * u + lane;
                if (j >= ke) continue;
                double x, y, ri, wgt, ex, ey, c, rho;
                cam_norm(R, (double)pp[u].x, (double)pp[u].y, (double)pp[u].z, x, y, ri);
                eval_obs(ck, x, y, zz[u], delta, ex, ey, c, rho, wgt);
                part += rho;
                if (store_chi) chik[j] = c;
                if (lin) {
                    double A[12], wA[12];
                    jac_norm(x, y, ri, A);
                    const double l0 = wgt * ck.fx2, l1 = wgt * ck.fy2;
#pragma unroll
                    for (int i = 0; i < 6; ++i) { wA[i] = l0 * A[i]; wA[6 + i] = l1 * A[6 + i]; }
                    int idx = 0;
#pragma unroll
                    for (int r = 0; r < 6; ++r)
#pragma unroll
                        for (int cc = r; cc < 6; ++cc) { acc[idx] = a_fma_pair(wA, r, A, cc, acc[idx]); ++idx; }
#pragma unroll
                    for (int r = 0; r < 6; ++r) acc[21 + r] = a_fma2(wA, r, -ex, -ey, acc[21 + r]);
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) { pp[u] = pn[u]; zz[u] = zn[u]; }
        }
        const double total = wave_sum(part);
        if (lin && mine) {
            wave_reduce_scatter<27>(acc, lane);
            if (slot27 >= 0 && slot27 < 21) {
                int r = 0, rem = slot27;
                while (rem >= 6 - r) { rem -= 6 - r; ++r; }
                const int cc = r + rem;
                sm.H[buf][wave][6 * r + cc] = acc[0]; sm.H[buf][wave][6 * cc + r] = acc[0];
            } else if (slot27 >= 21) sm.g[buf][wave][slot27 - 21] = acc[0];
        }
        return total;
    };
    auto sum_kf = [&](const double* v) -> double { // keyframe order: the same sum in every wave
        double t = 0;
        for (int k = 0; k < nk; ++k) t += v[k];
        return t;
    };
    vslam_lm_stats* st = a.stats ? a.stats + w : nullptr;
    double lambda = 0, ni = 2, currentChi = 0;
    int pc = 0; // which of sm.T / sm.R [0], [1] holds the current pose; sm.R[2] = the last evaluated trial (g2o's edge errors reflect it)
    int cur = 0, it = 0, total_trials = 0, pb = 0, fb = 0;
    if (iters > 0) { // the initial state goes through the same evaluation + linearisation
        const double c = evaluate(sm.R[pc][wave], true, cur, false);
        if (lane == 0 && mine) sm.part[pb][wave] = c;
        __syncthreads();
        currentChi = sum_kf(sm.part[pb]);
        pb ^= 1;
        double md = 0;
        for (int k = 0; k < nk; ++k)
#pragma unroll
            for (int d = 0; d < 6; ++d) md = fmax(md, fabs(sm.H[cur][k][7 * d]));
        lambda = 1e-5 * md;
        if (st && tid == 0) st->chi2_init = currentChi;
    }
    POH(1);
    for (it = 0; it < iters; ++it) {
        double rho_gain = 0;
        int qmax = 0;
        bool again = true;
        while (again) {
            double x[6] = {0, 0, 0, 0, 0, 0};
            if (mine) {
                double Hk[36], gk[6];
#pragma unroll
                for (int i = 0; i < 36; ++i) Hk[i] = sm.H[cur][wave][i];
#pragma unroll
                for (int i = 0; i < 6; ++i) gk[i] = sm.g[cur][wave][i];
                if (!chol6_solve(Hk, lambda, gk, x) && lane == 0) sm.flag[fb] = 1;
            }
            POH(2);
            __syncthreads();
            POH(3);
            const bool ok2 = sm.flag[fb] == 0;
            if (tid == 0) sm.flag[fb ^ 1] = 0; // (last read before the previous trial's second barrier, next written after this trial's)
            fb ^= 1;
            if (!ok2) {
#pragma unroll
                for (int i = 0; i < 6; ++i) x[i] = 0;
            }
            {
                double E[7], Tc[7], Tt[7], Rtt[12];
#pragma unroll
                for (int i = 0; i < 7; ++i) Tc[i] = sm.T[pc][wave][i];
                se3::exp(x, E);
                se3::mul(E, Tc, Tt);
                expand_pose(Tt, Rtt);
                __builtin_amdgcn_wave_barrier();
                if (lane == 0) {
#pragma unroll
                    for (int i = 0; i < 7; ++i) sm.T[pc ^ 1][wave][i] = Tt[i];
#pragma unroll
                    for (int i = 0; i < 12; ++i) { sm.R[pc ^ 1][wave][i] = Rtt[i]; sm.R[2][wave][i] = Rtt[i]; }
                }
                __builtin_amdgcn_wave_barrier();
            }
            POH(4);
            const double c = evaluate(sm.R[pc ^ 1][wave], true, cur ^ 1, false);
            POH(5);
            double sc = 0;
            if (mine) {
#pragma unroll
                for (int i = 0; i < 6; ++i) sc += x[i] * (lambda * x[i] + sm.g[cur][wave][i]);
            }
            if (lane == 0 && mine) { sm.part[pb][wave] = c; sm.spart[pb][wave] = sc; }
            __syncthreads();
            POH(6);
            double tempChi = sum_kf(sm.part[pb]);
            const double scale = sum_kf(sm.spart[pb]) + 1e-3;
            pb ^= 1;
            if (!ok2) tempChi = 1.7976931348623157e308;
            rho_gain = (currentChi - tempChi) / scale;
            if (rho_gain > 0 && isfinite(tempChi)) {
                double alpha = 1. - pow(2 * rho_gain - 1, 3);
                alpha = fmin(alpha, 2. / 3.);
                lambda *= fmax(1. / 3., alpha);
                ni = 2;
                currentChi = tempChi;
                pc ^= 1;
                cur ^= 1;
            } else {
                lambda *= ni;
                ni *= 2;
            }
            ++qmax;
            again = (rho_gain < 0) && qmax < 10;
        }
        total_trials += qmax;
        if (st && tid == 0 && it < VSLAM_LM_MAX_ITERS) { st->chi2_iter[it] = currentChi; st->lambda_iter[it] = lambda; st->trials_iter[it] = qmax; }
        if (qmax == 10 || rho_gain == 0) { ++it; break; }
    }
    if (st && tid == 0) { st->iterations = it; st->total_trials = total_trials; st->chi2_final = currentChi; st->lambda_final = lambda; }
    POH(7);
    // ---- chi2 of every active edge at the last evaluated state, then the adaptive threshold (optimization.cpp:224-252)
    evaluate(sm.R[2][wave], false, 0, true);
    double th = delta; // :154: chi2_th is both the Huber delta and the initial classification threshold
    int cb = 0;
    for (int iteration = 0; iteration < 5; ++iteration) {
        int cin = 0, cout = 0;
        for (int j = kb + lane; j < ke; j += 64) { if (chik[j] > th) ++cout; else ++cin; }
        for (int o = 32; o > 0; o >>= 1) { cin += __shfl_xor(cin, o); cout += __shfl_xor(cout, o); }
        if (lane == 0) { sm.cin[cb][wave] = mine ? cin : 0; sm.cout[cb][wave] = mine ? cout : 0; }
        __syncthreads();
        int tin = 0, tout = 0;
        for (int k = 0; k < nk; ++k) { tin += sm.cin[cb][k]; tout += sm.cout[cb][k]; }
        cb ^= 1;
        const double ratio = (double)tin / (double)(tin + tout);
        if (ratio > 0.5) break; // uniform
        th *= 2;
    }
    if (ka.want_chi2) { // chi2 in the caller's edge order, 0 for the edges of excluded landmarks (the flags are still the pass's input here)
        double* chi2 = a.chi2 + e0;
        for (int e = tid; e < ne; e += kPoBlock) if (!inl[lmi[e]]) chi2[e] = 0.0;
        for (int j = kb + lane; j < ke; j += 64) chi2[list[j]] = chik[j];
    }
    __syncthreads();
    // the last edge of a landmark decides its flag (:254-266, ascending edge order; the edges are sorted by landmark)
    for (int j = kb + lane; j < ke; j += 64) {
        const int e = list[j];
        if (e == ne - 1 || lmi[e + 1] != lmi[e]) inl[lmi[e]] = !(chik[j] > th);
    }
    if (tid == 0 && a.chi2_thr) a.chi2_thr[w] = th;
    if (update_poses && mine && lane < 7) a.T[((size_t)w * a.n_kf + wave) * 7 + lane] = sm.T[pc][wave][lane];
    POH(8);
    (void)ntot;
#undef POH
}

// (LmScratch, owned by the context and grown on demand, is declared in vslam_internal.h)
static int ensure(void** p, size_t* have, size_t need) {
    if (*have >= need) return VSLAM_OK;
    if (*p) hipFree(*p);
    *p = nullptr; *have = 0;
    if (hipMalloc(p, need) != hipSuccess) { set_error("LM scratch hipMalloc(%zu) failed", need); return VSLAM_ERR_HIP; }
    *have = need;
    return VSLAM_OK;
}

static int carve(LmScratch& g_lm, LmKernelArgs& ka, size_t total_lm, size_t total_edge, int n_windows, bool with_lm, hipStream_t stream) {
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t need = 0;
    const size_t o_P = need; need += al(total_lm * 3 * 8);
    const size_t o_Pt = need; need += al(total_lm * 3 * 8);
    const size_t o_Hll = need; need += al(total_lm * 6 * 8);
    const size_t o_bl = need; need += al(total_lm * 3 * 8);
    const size_t o_Di = need; need += al(total_lm * 6 * 8);
    const size_t o_lin = need; need += al(total_edge * kLin * 8);
    const size_t o_lmptr = need; need += al((total_lm + n_windows + 1) * 4);
    const size_t o_kfptr = need; need += al((size_t)n_windows * (kMaxKf + 1) * 4);
    const size_t o_kfe = need; need += al(total_edge * 4);
    const size_t o_pp = need; need += al((size_t)n_windows * (kMaxPairs + 1) * 4);
    const size_t o_hits = need; need += with_lm ? al(total_edge * kHitsPerEdge * 8) : 256;
    const size_t o_act = need; need += al(total_lm);
    const size_t o_kpos = need; need += al(total_edge * 4);
    const size_t o_st = need; need += al((size_t)n_windows * 4);
    const size_t o_chi = need; need += al(total_edge * 8);
    const size_t o_chik = need; need += al(total_edge * 8);
    const size_t o_uvk = need; need += al(total_edge * 8);
    const size_t o_suv = need; need += with_lm ? al(total_lm * kLmSlots * 8) : 256;
    const size_t o_skf = need; need += with_lm ? al(total_lm * kLmSlots) : 256;
    const size_t o_lcnt = need; need += al(total_lm);
    const size_t o_ord = need; need += al((size_t)n_windows * 4);
    const size_t o_pass = need; need += al((size_t)n_windows * 4);
    const size_t o_defer = need; need += al((size_t)n_windows * 4);
    // hipFree/hipMalloc are synchronising; growth only happens on the first call of a given size
    if (g_lm.bytes < need) { hipStreamSynchronize(stream); int rc = ensure(&g_lm.buf, &g_lm.bytes, need); if (rc) return rc; }
    uint8_t* base = (uint8_t*)g_lm.buf;
    ka.a.P = (double*)(base + o_P); ka.a.Ptrial = (double*)(base + o_Pt); ka.a.Hll = (double*)(base + o_Hll);
    ka.a.bl = (double*)(base + o_bl); ka.a.Dinv = (double*)(base + o_Di);
    ka.a.lin = (double*)(base + o_lin); ka.a.lm_ptr = (int32_t*)(base + o_lmptr); ka.a.kf_ptr = (int32_t*)(base + o_kfptr);
    ka.a.kf_edges = (int32_t*)(base + o_kfe); ka.a.pair_ptr = (int32_t*)(base + o_pp); ka.a.pair_hits = (int32_t*)(base + o_hits);
    ka.act = base + o_act; ka.kf_pos = (int32_t*)(base + o_kpos); ka.status = (int32_t*)(base + o_st);
    g_lm.status = ka.status; g_lm.status_n = n_windows;
    if (!ka.a.chi2) ka.a.chi2 = (double*)(base + o_chi);
    ka.chi2k = (double*)(base + o_chik); ka.uvk = (float*)(base + o_uvk);
    ka.slot_uv = (float*)(base + o_suv); ka.slot_kf = base + o_skf; ka.lcnt = base + o_lcnt;
    ka.order = (const int32_t*)(base + o_ord); // (filled by lm_order_kernel when the launcher wants it; cleared to null otherwise)
    ka.passes = (int32_t*)(base + o_pass); g_lm.passes = ka.passes;
    g_lm.defer = (int32_t*)(base + o_defer);
    return VSLAM_OK;
}

// Largest window first.  A batch has more windows than the chip has CUs (one workgroup per CU), a window's time grows with its edge count, and
// the hardware hands the next workgroup to the first CU that frees up: in batch order a big window can be the last one started and the
// whole launch waits for it while 255 CUs idle; in descending order the last ones started are the smallest (longest-processing-time-first list
// scheduling).  One workgroup sorts (edge count descending, window index ascending) keys in LDS; which workgroup computes a window changes nothing
// in its result.
#ifndef VSLAM_LM_LPT
#define VSLAM_LM_LPT 1
#endif
__global__ void lm_fill_kernel(int32_t* __restrict__ p, int n, int v) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v; }
constexpr int kOrderCap = 4096;
__global__ __launch_bounds__(1024) void lm_order_kernel(const int32_t* __restrict__ edge_off, int n, int32_t* __restrict__ order) {
    __shared__ unsigned long long key[kOrderCap];
    int np2 = 1;
    while (np2 < n) np2 <<= 1;
    for (int i = threadIdx.x; i < np2; i += 1024)
        key[i] = i < n ? ((unsigned long long)(0xFFFFFFFFu - (unsigned)max(edge_off[i + 1] - edge_off[i], 0)) << 32) | (unsigned)i : ~0ull;
    for (int k = 2; k <= np2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            for (int t = threadIdx.x; t < np2 / 2; t += 1024) {
                const int lo = ((t / j) * 2 * j) + (t % j), hi = lo + j;
                const bool up = (lo & k) == 0;
                const unsigned long long x = key[lo], y = key[hi];
                if ((x > y) == up) { key[lo] = y; key[hi] = x; }
            }
        }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 1024) order[i] = (int32_t)(key[i] & 0xFFFFFFFFu);
}

int launch_lm_windows(const LmWindowArgs& a, int schedule, int mode, int iters, int update_poses, int update_lms, LmScratch* scratch,
                      hipStream_t stream) {
    const size_t total_lm = a.total_lm, total_edge = a.total_edge;
    if (a.n_windows <= 0) return VSLAM_OK;
    if (a.n_kf <= 0 || a.n_kf > kMaxKf) { set_error("n_kf %d out of range (1..%d)", a.n_kf, kMaxKf); return VSLAM_ERR_ARG; }
    LmKernelArgs ka;
    memset(&ka, 0, sizeof(ka));
    ka.a = a;
    ka.want_chi2 = a.chi2 != nullptr;
    ka.dinv_lds = kDinvLds;
    const size_t dyn_lds = (size_t)kDinvLds * 6 * sizeof(double);
    if (!scratch->lds_opt_in) { // more than 64 KB of dynamic LDS needs the opt-in (once per context, i.e. per device)
        VS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&lm_window_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_lds));
        VS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&lm_window_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_lds));
        scratch->lds_opt_in = true;
    }
    static long long* d_cyc = nullptr; static int cyc_n = 0;
    if (getenv("VSLAM_LM_PROFILE")) {
        if (cyc_n < a.n_windows) { if (d_cyc) hipFree(d_cyc); hipMalloc((void**)&d_cyc, sizeof(long long) * kDbgSlots * a.n_windows); cyc_n = a.n_windows; }
        hipMemsetAsync(d_cyc, 0, sizeof(long long) * kDbgSlots * a.n_windows, stream);
        ka.dbg_cycles = d_cyc;
    }
    int rc = carve(*scratch, ka, total_lm, total_edge, a.n_windows, true, stream);
    if (rc) return rc;
    // Tuning::ba_adaptive (default on): ONE launch runs a window's passes back to back, and a pass that flags nothing new is continued to the last pass's
    // 10 iterations instead of being repeated (see the kernel); 0: the three passes as three launches, every one of them for every window
    const bool adaptive = !(scratch->tune && scratch->tune->ba_adaptive == 0);
    const bool resident_on = (schedule || mode == 0) && !(scratch->tune && scratch->tune->ba_resident == 0);
    ProfScope prof__(stream, "lm_window_kernel", (schedule ? (adaptive ? 2 : 4) : 1) + (resident_on ? 1 : 0)); // (family name kept from rounds 1-4; the BA schedule's launches)
    if (VSLAM_LM_LPT && a.n_windows > 1 && a.n_windows <= kOrderCap)
        hipLaunchKernelGGL(lm_order_kernel, dim3(1), dim3(1024), 0, stream, a.edge_off, a.n_windows, const_cast<int32_t*>(ka.order));
    else ka.order = nullptr;
    // optimize_map passes: ba_resident_kernel takes every window whose landmark state fits the LDS of a CU (Tuning::ba_resident = 0: none) and marks
    // the others in `defer`; lm_window_kernel then runs with that list and returns at once for the windows that are done
    const bool resident = (schedule || mode == 0) && !(scratch->tune && scratch->tune->ba_resident == 0);
    scratch->defer_valid = resident;
    if (resident) {
        if (scratch->rs_dyn_bytes < 0) { int dev = 0; (void)hipGetDevice(&dev); scratch->rs_dyn_bytes = rs_dyn_lds_bytes(dev); }
        RsLaunch L;
        memset(&L, 0, sizeof(L));
        L.a = ka.a; L.uv_s = ka.uvk; L.epos = ka.kf_pos; L.tab = ka.a.Hll; L.xin = ka.a.Ptrial; L.Pbak = ka.a.P; L.Dc = ka.a.Dinv; L.blc = ka.a.bl;
        L.status = ka.status; L.passes = ka.passes; L.defer = scratch->defer; L.order = ka.order; L.dbg = getenv("VSLAM_RS_PROFILE") ? ka.dbg_cycles : nullptr;
        L.lanes = (scratch->tune && scratch->tune->ba_lanes > 0) ? scratch->tune->ba_lanes : 0;
        L.dyn_bytes = scratch->rs_dyn_bytes; L.schedule = schedule; L.adaptive = adaptive ? 1 : 0; L.iters = iters; L.update_poses = update_poses; L.update_lms = update_lms;
        L.opt_in_done = scratch->rs_opt_in; L.want_chi2 = ka.want_chi2;
        L.dense_to_general = !(scratch->tune && scratch->tune->ba_resident == 1);
        if (schedule && !adaptive) hipLaunchKernelGGL(lm_fill_kernel, dim3((a.n_windows + 255) / 256), dim3(256), 0, stream, ka.passes, a.n_windows, 3);
        rc = launch_ba_resident(L, stream);
        if (rc) return rc;
        scratch->rs_opt_in = true;
        ka.defer = scratch->defer;
        if (L.dbg) { // tuning aid (VSLAM_LM_PROFILE=1 VSLAM_RS_PROFILE=1): phase clocks of ba_resident_kernel, thread 0 of every window
            hipStreamSynchronize(stream);
            std::vector<long long> h(kDbgSlots * (size_t)a.n_windows);
            hipMemcpy(h.data(), ka.dbg_cycles, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
            static const char* rs_names[16] = {"setup", "pass init", "boot evaluation", "lambda-init pass", "linearise", "convert+cholesky+solve", "pose update+backsub+trial eval",
                                               "(loop tail)", "classification", "write-back", "lin: row open", "lin: landmark blocks", "lin: diag pair (row-wise)", "lin: off-diag pairs (row-wise)", "lin: singles chunks", "(classification: row loop)"};
            double tot = 0;
            for (int i = 0; i < 16; ++i) { double sum = 0; for (int w = 0; w < a.n_windows; ++w) sum += (double)h[16 * (size_t)w + i]; sum /= a.n_windows; if (i < 10) tot += sum; fprintf(stderr, "  [rs profile] %-32s %10.0f ticks/window\n", rs_names[i], sum); }
            fprintf(stderr, "  [rs profile] total %.0f (clock64 ticks)\n", tot);
            hipMemsetAsync(ka.dbg_cycles, 0, sizeof(long long) * kDbgSlots * a.n_windows, stream);
        }
    }
    if (schedule) {
        // run_vslam.cpp:58-71: optimize_map(5) x2 without write-back, optimize_map(10) writing poses, optimize_pose_only(10)
        if (adaptive) {
            hipLaunchKernelGGL((lm_window_kernel<false, true>), dim3(a.n_windows), dim3(kLmBlock), dyn_lds, stream, ka, 0, 5, 0, 0, 1, 0);
        } else {
            if (!resident) hipLaunchKernelGGL(lm_fill_kernel, dim3((a.n_windows + 255) / 256), dim3(256), 0, stream, ka.passes, a.n_windows, 3);
            hipLaunchKernelGGL(lm_window_kernel<false>, dim3(a.n_windows), dim3(kLmBlock), dyn_lds, stream, ka, 0, 5, 0, 0, 1, 0);
            hipLaunchKernelGGL(lm_window_kernel<false>, dim3(a.n_windows), dim3(kLmBlock), dyn_lds, stream, ka, 0, 5, 0, 0, 1, 1); // (the landmark CSR of the first launch is still valid)
            hipLaunchKernelGGL(lm_window_kernel<false>, dim3(a.n_windows), dim3(kLmBlock), dyn_lds, stream, ka, 0, kSchedFinalIters, 1, 0, 1, 1);
        }
        // the pose-only pass: one wave per keyframe (pose_only_wave_kernel); Tuning::pose_only_window = 1 (tuning aid / cross-check test) keeps the window kernel
        ka.defer = nullptr; // (every window)
        const bool po_window = scratch->tune && scratch->tune->pose_only_window > 0;
        if (po_window) hipLaunchKernelGGL(lm_window_kernel<false>, dim3(a.n_windows), dim3(kLmBlock), dyn_lds, stream, ka, 1, 10, 1, 0, 1, resident ? 0 : 1); // (the landmark CSR exists only where lm_window_kernel ran the passes)
        else {
            if (ka.dbg_cycles && getenv("VSLAM_PO_PROFILE")) hipMemsetAsync(ka.dbg_cycles, 0, sizeof(long long) * kDbgSlots * a.n_windows, stream); // show only this pass
            hipLaunchKernelGGL(pose_only_wave_kernel, dim3(a.n_windows), dim3(kPoBlock), 0, stream, ka, 10, 1);
        }
    } else {
        hipLaunchKernelGGL(lm_window_kernel<false>, dim3(a.n_windows), dim3(kLmBlock), dyn_lds, stream, ka, mode, iters, update_poses, update_lms, 1, 0);
    }
    VS_HIP(hipGetLastError());
    if (ka.dbg_cycles) {
        hipStreamSynchronize(stream);
        std::vector<long long> h(kDbgSlots * (size_t)a.n_windows);
        hipMemcpy(h.data(), ka.dbg_cycles, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
        static const char* names[kDbgSlots] = {"setup", "eval+lin", "lm blocks", "pose blocks", "lambda/Dinv", "bs", "schur", "cholesky (rest)", "solve (rest)", "update+scale", "eval trial", "loop tail", "classify+wb", "(setup: csr)", "(setup: kf-major)", "schur: item prologue",
                                               "chol: diag+rows", "chol: barrier 1", "chol: trailing update", "chol: barrier 2", "solve: barrier", "solve: back-subst", "schur: rows", "schur: reduce+store"};
        double tot = 0;
        for (int i = 0; i < kDbgSlots; ++i) { double s = 0; for (int w = 0; w < a.n_windows; ++w) s += (double)h[kDbgSlots * (size_t)w + i]; s /= a.n_windows; if (i < 13 || i >= 16) tot += s; fprintf(stderr, "  [lm profile] %-20s %10.0f ticks/window\n", names[i], s); }
        fprintf(stderr, "  [lm profile] total %.0f cycles (clock64 = shader clock, thread 0 of every window; setup sub-splits not included)\n", tot);
    }
    return VSLAM_OK;
}

int lm_fetch_deferred(const LmScratch* scratch, int n_windows, int32_t* h_defer, hipStream_t stream) {
    if (!scratch->buf || !scratch->defer || !h_defer || n_windows > scratch->status_n) return VSLAM_ERR_ARG;
    if (!scratch->defer_valid) { for (int w = 0; w < n_windows; ++w) h_defer[w] = 1; return VSLAM_OK; } // the last launch did not involve ba_resident_kernel: every window on lm_window_kernel
    VS_HIP(hipMemcpyAsync(h_defer, scratch->defer, sizeof(int32_t) * n_windows, hipMemcpyDeviceToHost, stream));
    VS_HIP(hipStreamSynchronize(stream));
    for (int w = 0; w < n_windows; ++w) h_defer[w] &= 1;
    return VSLAM_OK;
}

int lm_fetch_passes(const LmScratch* scratch, int n_windows, int32_t* h_passes, hipStream_t stream) {
    if (!scratch->buf || !scratch->passes || !h_passes || n_windows > scratch->status_n) return VSLAM_ERR_ARG;
    VS_HIP(hipMemcpyAsync(h_passes, scratch->passes, sizeof(int32_t) * n_windows, hipMemcpyDeviceToHost, stream));
    VS_HIP(hipStreamSynchronize(stream));
    return VSLAM_OK;
}

int lm_fetch_status(const LmScratch* scratch, int n_windows, int32_t* h_status, hipStream_t stream) {
    const LmScratch& g_lm = *scratch;
    if (!g_lm.buf || !h_status) return VSLAM_ERR_ARG;
    // the status words of the most recent launch live at a fixed offset that carve() recorded
    VS_HIP(hipMemcpyAsync(h_status, g_lm.status, sizeof(int32_t) * n_windows, hipMemcpyDeviceToHost, stream));
    VS_HIP(hipStreamSynchronize(stream));
    return VSLAM_OK;
}

// diagnostic of rows A10 / A11: one lane per observation runs the device functions every LM kernel linearises with (cam_norm, eval_obs,
// jac_norm, jac_point_norm) and scales the normalised-coordinate factors back to pixels: A = diag(fx, fy) At, B = diag(fx, fy) Bt, e = diag(fx, fy) en
__global__ __launch_bounds__(256) void edge_jacobian_kernel(int n, const float* __restrict__ xyz, const float* __restrict__ uv, const double* __restrict__ T, double K0, double K1,
                                                           double K2, double K3, double delta, double* __restrict__ err, double* __restrict__ Jp, double* __restrict__ Jl,
                                                           double* __restrict__ chi2, double* __restrict__ hw) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double K[4] = {K0, K1, K2, K3};
    const CamK ck = make_camk(K);
    double Tl[7], Rt[12];
#pragma unroll
    for (int c = 0; c < 7; ++c) Tl[c] = T[c];
    expand_pose(Tl, Rt);
    double x, y, rho, enx, eny, c, rob, wg, A[12], B[6];
    cam_norm(Rt, (double)xyz[3 * i], (double)xyz[3 * i + 1], (double)xyz[3 * i + 2], x, y, rho);
    eval_obs(ck, x, y, reinterpret_cast<const float2*>(uv)[i], delta, enx, eny, c, rob, wg);
    jac_norm(x, y, rho, A);
    jac_point_norm(x, y, rho, Rt, B);
    if (err) { err[2 * i] = ck.fx * enx; err[2 * i + 1] = ck.fy * eny; }
    if (Jp)
#pragma unroll
        for (int a = 0; a < 6; ++a) { Jp[12 * i + a] = ck.fx * A[a]; Jp[12 * i + 6 + a] = ck.fy * A[6 + a]; }
    if (Jl)
#pragma unroll
        for (int a = 0; a < 3; ++a) { Jl[6 * i + a] = ck.fx * B[a]; Jl[6 * i + 3 + a] = ck.fy * B[3 + a]; }
    if (chi2) chi2[i] = c;
    if (hw) hw[i] = wg;
}
int launch_edge_jacobians(int n, const float* d_xyz, const float* d_uv, const double* d_T, const double K[4], double delta, double* d_err, double* d_Jp, double* d_Jl,
                          double* d_chi2, double* d_hw, hipStream_t stream) {
    if (n <= 0) return VSLAM_OK;
    hipLaunchKernelGGL(edge_jacobian_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, n, d_xyz, d_uv, d_T, K[0], K[1], K[2], K[3], delta, d_err, d_Jp, d_Jl, d_chi2, d_hw);
    VS_HIP(hipGetLastError());
    return VSLAM_OK;
}

int launch_pnp(const PnpArgs& p, LmScratch* scratch, hipStream_t stream) {
    if (p.B <= 0) return VSLAM_OK;
    // one wave per problem unless the caller knows the problems are large (n_hint points: the window kernel's 512 lanes pay from ~1000
    // points on); Tuning::pnp_window = 1 (tuning aid / cross-check test) forces the window kernel
    const bool force_window = scratch->tune && scratch->tune->pnp_window > 0;
    if (!force_window && p.n_hint <= 1024) {
        ProfScope prof__(stream, "pnp_wave_kernel");
        hipLaunchKernelGGL(pnp_wave_kernel, dim3(p.B), dim3(64), 0, stream, p.xyz, p.uv, p.n, p.capacity, p.T, p.iters, p.K[0], p.K[1], p.K[2], p.K[3],
                           p.huber_delta, p.reproj_thr * p.reproj_thr, p.inlier, p.n_inliers, p.stats);
        VS_HIP(hipGetLastError());
        return VSLAM_OK;
    }
    LmKernelArgs ka;
    memset(&ka, 0, sizeof(ka));
    ka.a.n_windows = p.B; ka.a.n_kf = 1;
    ka.a.T = p.T; ka.a.xyz = const_cast<float*>(p.xyz); ka.a.uv = p.uv; ka.a.stats = p.stats;
    ka.a.K[0] = p.K[0]; ka.a.K[1] = p.K[1]; ka.a.K[2] = p.K[2]; ka.a.K[3] = p.K[3];
    ka.a.huber_delta = p.huber_delta;
    ka.pnp_n = p.n; ka.capacity = p.capacity;
    const size_t tot = (size_t)p.B * p.capacity;
    int rc = carve(*scratch, ka, tot, tot, p.B, false, stream);
    if (rc) return rc;
    ka.order = nullptr; // (problem b runs on workgroup b)
    ProfScope prof__(stream, "lm_window_kernel<pnp>", 2);
    hipLaunchKernelGGL(lm_window_kernel<true>, dim3(p.B), dim3(kLmBlock), 0, stream, ka, 1, p.iters, 1, 0, 0, 0);
    hipLaunchKernelGGL(pnp_inlier_kernel, dim3(p.B), dim3(256), 0, stream, p.xyz, p.uv, p.n, p.capacity, p.T, p.K[0], p.K[1], p.K[2], p.K[3],
                       p.reproj_thr * p.reproj_thr, p.inlier, p.n_inliers);
    VS_HIP(hipGetLastError());
    return VSLAM_OK;
}

} // namespace vslam
