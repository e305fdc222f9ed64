// track_kernels.hip -- device-side graph construction for the BA half of a throughput-mode step.
//
// What it replaces: the landmark / observation bookkeeping of VO::insert_key_frame
// (/root/reference/src/stereo_visual_slam_main/visual_odometry.cpp:363-424: a tracked feature adds an observation to the landmark of
// the feature it was matched to, every other keypoint with a valid depth creates a landmark, a landmark whose depth was unreliable
// takes the position of the first later observation with a reliable depth, :391-401) and the graph build of optimize_map /
// optimize_pose_only (optimization.cpp:127-214, :303-361: poses = the keyframes of the window, landmarks with >= 1 observation, one
// edge per observation) -- for a batch of B CONSECUTIVE keyframes whose front-end results are already in device memory.  Window b is the
// map right after keyframe b was inserted: keyframes [max(0, b - n_kf + 1), b] (Map::num_keyframes_ = 10, map.hpp:22), every landmark
// observed by one of them, positions and reliable_depth_ as of time b.  Poses: the pose stage's relative poses chained from frame 0.
//
// Throughput-mode simplifications (stated in DESIGN.md): every frame is a keyframe, windows are independent (is_inlier = 1 on entry:
// the chi2 classification of window b - 1 does not feed window b), the sliding window replaces the distance-based culling of
// Map::remove_keyframe (map.cpp:48-130).
//
// gfx950 mapping: the reference walks std::unordered_map<id, Landmark> with per-landmark observation vectors; here a track is a chain
// of (frame, keypoint) nodes linked by two flat int32 tables pred / succ (B x kp_capacity) filled by one scatter pass per frame pair,
// every keypoint slot of the batch is a thread, and a window is one workgroup that ranks the chain HEADS inside its frames with ballots
// -- landmark-sorted, window-local edge lists come out directly, nothing is sorted.  Landmark order inside a window: by observation
// count, then by the head's (frame, keypoint).  The optimiser accepts any landmark order; THIS one makes the 64 consecutive landmarks a
// wave of its landmark-wise phases owns homogeneous (most landmarks of a real sequence are seen once, a few in all ten keyframes: in
// creation order every wave ran ten observation rounds for an average of 1.3 useful ones).
// All kernels are byte / index work on < 25 MB of tables per 256 frames: bound by launch latency, not by bandwidth.
#include "vslam_internal.h"

#include "se3_device.h"

namespace vslam {

struct TrackDims { int B, kp_cap, lr_cap, match_cap, pnp_cap, n_kf; };

// ---- per frame: keypoint -> L/R match table; chain tables cleared
__global__ __launch_bounds__(256) void track_init_kernel(TrackDims d, const vslam_dmatch* __restrict__ d_lr, const int32_t* __restrict__ d_nlr,
                                                        int32_t* __restrict__ kp2lr, int32_t* __restrict__ pred, int32_t* __restrict__ succ, int32_t* __restrict__ cand) {
    const int f = blockIdx.x, tid = threadIdx.x;
    int32_t* k2 = kp2lr + (size_t)f * d.kp_cap;
    for (int i = tid; i < d.kp_cap; i += 256) { k2[i] = -1; pred[(size_t)f * d.kp_cap + i] = -1; succ[(size_t)f * d.kp_cap + i] = -1; cand[(size_t)f * d.kp_cap + i] = -1; }
    __syncthreads();
    const int nlr = min(max(d_nlr[f], 0), d.lr_cap);
    const vslam_dmatch* lr = d_lr + (size_t)f * d.lr_cap;
    for (int m = tid; m < nlr; m += 256) {
        const int q = lr[m].queryIdx;
        if (q >= 0 && q < d.kp_cap) k2[q] = m;
    }
}

// ---- global poses: G[0] = identity, G[f] = T_rel[f - 1] o G[f - 1]; inclusive scan of SE3 products (Hillis-Steele in LDS, chunks of 256
// frames chained through a carry).  SE3 composition is associative; the scan's grouping differs from a sequential chain only in rounding.
__global__ __launch_bounds__(256) void track_pose_chain_kernel(int B, const double* __restrict__ T_rel, double* __restrict__ G) {
    __shared__ double buf[2][256][7];
    __shared__ double carry[7];
    const int tid = threadIdx.x;
    if (tid == 0) { carry[0] = carry[1] = carry[2] = 0; carry[3] = 1; carry[4] = carry[5] = carry[6] = 0; }
    for (int base = 0; base < B; base += 256) {
        const int f = base + tid;
        double X[7] = {0, 0, 0, 1, 0, 0, 0};
        if (f > 0 && f < B)
#pragma unroll
            for (int i = 0; i < 7; ++i) X[i] = T_rel[(size_t)(f - 1) * 7 + i];
        int cur = 0;
#pragma unroll
        for (int i = 0; i < 7; ++i) buf[0][tid][i] = X[i];
        __syncthreads();
        for (int dd = 1; dd < 256; dd <<= 1) {
            double Y[7];
            if (tid >= dd) se3::mul(buf[cur][tid], buf[cur][tid - dd], Y); // later frames on the left
            else
#pragma unroll
                for (int i = 0; i < 7; ++i) Y[i] = buf[cur][tid][i];
#pragma unroll
            for (int i = 0; i < 7; ++i) buf[cur ^ 1][tid][i] = Y[i];
            cur ^= 1;
            __syncthreads();
        }
        double Gf[7];
        se3::mul(buf[cur][tid], carry, Gf);
        if (f < B)
#pragma unroll
            for (int i = 0; i < 7; ++i) G[(size_t)f * 7 + i] = Gf[i];
        __syncthreads();
        if (tid == 255)
#pragma unroll
            for (int i = 0; i < 7; ++i) carry[i] = Gf[i];
        __syncthreads();
    }
}

// ---- per frame pair (i -> i + 1): every frame-to-frame match becomes a CANDIDATE link q -> t.  Input j of the pose stage is the j-th match
// whose query keypoint owns a valid depth (the compaction of build_pnp_inputs_kernel, geom_kernels.hip, repeated with the same ballot ranks):
// such a candidate carries the pose stage's inlier flag (the reference erases the outliers of motion_estimation from the frame, :306).  A
// candidate whose query keypoint has no depth of its own is decided by the walk below (track_rule 1) or never a link (track_rule 0).
constexpr int kCandDepth = 1 << 20, kCandInlier = 1 << 21, kCandIndex = 0xFFFF; // cand word of slot t: q | flags (kp_capacity <= 65536), -1: no match reaches it
__global__ __launch_bounds__(256) void track_link_kernel(TrackDims d, const vslam_dmatch* __restrict__ d_f2f, const int32_t* __restrict__ d_nf2f,
                                                        const uint8_t* __restrict__ d_valid, const uint8_t* __restrict__ d_inl,
                                                        const int32_t* __restrict__ kp2lr, int32_t* __restrict__ cand, int32_t* __restrict__ succ) {
    const int it = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ int s_tot[4];
    const int nm = min(max(d_nf2f[it], 0), d.match_cap);
    const vslam_dmatch* m = d_f2f + (size_t)it * d.match_cap;
    const int32_t* k2 = kp2lr + (size_t)it * d.kp_cap;
    int written = 0;
    for (int base = 0; base < nm; base += 256) {
        const int k = base + tid;
        bool ok = false, in_range = false; int q = -1, t = -1;
        if (k < nm) {
            q = m[k].queryIdx; t = m[k].trainIdx;
            in_range = q >= 0 && q < d.kp_cap && t >= 0 && t < d.kp_cap;
            if (in_range) { const int li = k2[q]; ok = li >= 0 && d_valid[(size_t)it * d.lr_cap + li] != 0; }
        }
        const unsigned long long mask = __ballot(ok);
        __syncthreads();
        if (lane == 0) s_tot[wave] = __popcll(mask);
        __syncthreads();
        int off = written;
        for (int w = 0; w < wave; ++w) off += s_tot[w];
        const int j = off + __popcll(mask & ((1ull << lane) - 1ull));
        if (in_range) { // (matches are one-to-one in query and train index: the matcher's cross-check)
            const bool inl = ok && j < d.pnp_cap && d_inl[(size_t)it * d.pnp_cap + j] != 0;
            cand[(size_t)(it + 1) * d.kp_cap + t] = q | (ok ? kCandDepth : 0) | (inl ? kCandInlier : 0);
            succ[(size_t)it * d.kp_cap + q] = t;
        }
        written += s_tot[0] + s_tot[1] + s_tot[2] + s_tot[3];
    }
}

constexpr int kCarryCode = -2;
__device__ inline int carry_flags(const float* carry, int slot) { return carry ? (int)carry[4 * slot + 3] : 0; }

// the landmark position a chain node stands for: the point of node `src` (= first reliable node of the chain, else its root) in the world of G,
// or the carried position when the chain's source lies before the batch
__device__ inline void landmark_position(const TrackDims& d, int src, const int32_t* __restrict__ kp2lr, const float* __restrict__ d_xyz, const double* __restrict__ G,
                                         const float* __restrict__ carry, float out[3]) {
    if (src <= kCarryCode) { const int slot = kCarryCode - src; out[0] = carry[4 * slot]; out[1] = carry[4 * slot + 1]; out[2] = carry[4 * slot + 2]; return; }
    const int sf = src / d.kp_cap, si = src - sf * d.kp_cap;
    const int mm = kp2lr[(size_t)sf * d.kp_cap + si];
    const float* pc = d_xyz + 3 * ((size_t)sf * d.lr_cap + mm);
    double Gi[7], pw[3];
    const double p[3] = {(double)pc[0], (double)pc[1], (double)pc[2]};
    se3::inverse(G + (size_t)sf * 7, Gi);
    se3::act(Gi, p, pw);
    out[0] = (float)pw[0]; out[1] = (float)pw[1]; out[2] = (float)pw[2];
}

// PnPRansac's inlier test (visual_odometry.cpp:277, reprojection error <= 4 px; the contract of the pose stage's own flags) on a landmark's
// map position seen through the chained pose of the current frame
struct TrackCam { double fx, fy, cx, cy, thr2; int track_rule; };
__device__ inline bool reprojects_within(const float pos[3], const double* __restrict__ T, const vslam_keypoint* __restrict__ kp, const TrackCam& cam) {
    const double pw[3] = {(double)pos[0], (double)pos[1], (double)pos[2]};
    double pc[3];
    se3::act(T, pw, pc);
    const double du = (double)kp->x - (cam.fx * pc[0] / pc[2] + cam.cx), dv = (double)kp->y - (cam.fy * pc[1] / pc[2] + cam.cy);
    const double c = du * du + dv * dv;
    return isfinite(c) && c <= cam.thr2;
}

// ---- the tracks.  The candidate links form disjoint PATHS through the batch (a slot has at most one candidate in and one out); the thread of
// a path's first slot (no candidate reaches it, or it lies in the batch's first frame) walks the path forward in time, carrying the landmark the
// current slot stands for, and decides every candidate the way VO::tracking / motion_estimation would (visual_odometry.cpp:568-599, :260-306):
//   a slot with a valid depth of its own that no track reaches creates a landmark (:381-421: root = itself);
//   a candidate OUT of a slot that is a feature continues the track when the pose stage kept it (the slot owns a depth: it was an input), or -- [r6],
//   track_rule 1: the reference's query set is every feature of the last frame (:568-574), depth or not -- when the landmark's map position
//   (the creation point, or the first reliable one, :391-401, as of the last frame) reprojects within the pose stage's 4 px through the current frame's pose;
//   a candidate out of a slot that is no feature is nothing.
// Per slot it leaves: root (where the landmark was created), relsrc (the FIRST node of the chain, up to this one, with a reliable depth; -1: none yet),
// pred / succ (the links that hold).  A chunk of a longer sequence (carry: vslam_tracks_in::d_carry_in): a track may reach a keypoint of the batch's
// first frame from BEFORE the batch.  Such a slot is a feature whatever its own depth, and the chain's root -- and its first reliable node, if the
// carry says one was seen -- lie upstream: both tables then hold the code  kCarryCode - slot  (< -1), which resolves to the carried position.
// Sequential per path (its length: a few frames for most, the batch at worst), parallel over the ~B x 1200 paths.
__global__ __launch_bounds__(256) void track_walk_kernel(TrackDims d, TrackCam cam, const vslam_keypoint* __restrict__ d_kps, const float* __restrict__ d_xyz,
                                                        const uint8_t* __restrict__ d_valid, const uint8_t* __restrict__ d_rel, const int32_t* __restrict__ kp2lr,
                                                        const int32_t* __restrict__ cand, int32_t* __restrict__ pred, int32_t* __restrict__ succ,
                                                        const double* __restrict__ G, const float* __restrict__ carry, int32_t* __restrict__ root,
                                                        int32_t* __restrict__ relsrc) {
    const int f0 = blockIdx.y, i0 = blockIdx.x * 256 + threadIdx.x;
    if (i0 >= d.kp_cap) return;
    if (f0 > 0 && cand[(size_t)f0 * d.kp_cap + i0] >= 0) return; // some earlier slot's walk passes through here
    int cf = f0, ci = i0;
    bool feat = false;
    int r = -1, first = -1; // the landmark of the current slot: root and first reliable node (codes as in the tables)
    if (f0 == 0) {
        const int cfl = carry_flags(carry, i0);
        if (cfl & 1) { feat = true; r = kCarryCode - i0; if (cfl & 2) first = r; }
    }
    int pos_src = 0x7FFFFFFF; float pos[3] = {0.f, 0.f, 0.f}; // map position of the landmark, computed when a candidate needs it
    for (;;) {
        const size_t c = (size_t)cf * d.kp_cap + ci;
        const int node = cf * d.kp_cap + ci;
        const int mm = kp2lr[c];
        const bool own3d = mm >= 0 && d_valid[(size_t)cf * d.lr_cap + mm] != 0;
        if (!feat && own3d) { feat = true; r = node; first = -1; }                                        // :403-418 a landmark is created here
        if (feat && first == -1 && own3d && d_rel[(size_t)cf * d.lr_cap + mm] != 0) first = node;           // :391-401 (or created reliable)
        root[c] = feat ? r : -1; relsrc[c] = feat ? first : -1;
        if (cf + 1 >= d.B) break;
        const int t = succ[c]; // (candidate)
        if (t < 0) break;
        const size_t cn = (size_t)(cf + 1) * d.kp_cap + t;
        const int cd = cand[cn];
        bool holds = false;
        if (feat) {
            if (cd & kCandDepth) holds = (cd & kCandInlier) != 0;
            else if (cam.track_rule) {
                const int src = first != -1 ? first : r;
                if (src != pos_src) { landmark_position(d, src, kp2lr, d_xyz, G, carry, pos); pos_src = src; }
                holds = reprojects_within(pos, G + (size_t)(cf + 1) * 7, d_kps + cn, cam);
            }
        }
        if (holds) pred[cn] = ci;
        else { succ[c] = -1; feat = false; r = -1; first = -1; }
        ++cf; ci = t;
    }
}

// ---- per slot, after the walk: what a window needs to know about it without walking: node?, has a predecessor?, successors left in its chain
__global__ __launch_bounds__(256) void track_info_kernel(TrackDims d, const int32_t* __restrict__ pred, const int32_t* __restrict__ succ, const int32_t* __restrict__ root,
                                                        const float* __restrict__ carry, int32_t* __restrict__ info) {
    const int f = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= d.kp_cap) return;
    const size_t at = (size_t)f * d.kp_cap + i;
    int p = pred[at];
    if (f == 0 && (carry_flags(carry, i) & 1)) p = 0; // (tracked from before the batch: a node with a predecessor; the index itself is never followed)
    const int r = root[at];
    int rem = 0;
    const bool node = r >= 0 || r <= kCarryCode;
    if (node) {
        int cf = f, ci = i;
        while (cf + 1 < d.B && rem < VSLAM_MAX_KF) {
            const int nx = succ[(size_t)cf * d.kp_cap + ci];
            if (nx < 0) break;
            ++cf; ci = nx; ++rem;
        }
    }
    info[at] = (node ? 1 : 0) | (p >= 0 ? 2 : 0) | (rem << 8);
}

// ---- carry-out for the chunk that starts at frame c of this batch: per keypoint slot of frame c, does a track reach it from frame c - 1, and
// what is that track's landmark position / reliable flag AS OF ITS NODE IN FRAME c - 1 (the state the next chunk's chain walk would have found upstream)
__global__ __launch_bounds__(256) void track_carry_out_kernel(TrackDims d, int c, const int32_t* __restrict__ kp2lr, const int32_t* __restrict__ pred,
                                                             const int32_t* __restrict__ root, const int32_t* __restrict__ relsrc, const float* __restrict__ d_xyz,
                                                             const double* __restrict__ G, const float* __restrict__ carry_in, float* __restrict__ carry_out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= d.kp_cap) return;
    float rec[4] = {0.f, 0.f, 0.f, 0.f};
    const int p = pred[(size_t)c * d.kp_cap + i];
    if (p >= 0) {
        const size_t pn = (size_t)(c - 1) * d.kp_cap + p;
        const int rs = relsrc[pn], src = (rs >= 0 || rs <= kCarryCode) ? rs : root[pn];
        landmark_position(d, src, kp2lr, d_xyz, G, carry_in, rec);
        rec[3] = (rs >= 0 || rs <= kCarryCode) ? 3.f : 1.f;
    }
    reinterpret_cast<float4*>(carry_out)[i] = make_float4(rec[0], rec[1], rec[2], rec[3]);
}

// A chain HEAD of window [s, b]: a node in frame s, or a node without predecessor (a landmark created inside the window).  Every
// landmark observed in the window has exactly one.  Returns the observations it has inside the window (0: not a head) -- from the
// slot's info word alone (track_chain_kernel), no chain walk.
__device__ inline int window_head_len(int info, int s, int b, int f) {
    if (!(info & 1) || (f != s && (info & 2))) return 0;
    return min((info >> 8) + 1, b - f + 1);
}

__device__ inline int block_sum_i32(int v, int* red /* 4 */) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// ---- per window: landmark and edge counts, and the landmarks per observation count (the bins of the emit pass)
constexpr int kHist = VSLAM_MAX_KF + 1;
__global__ __launch_bounds__(256) void window_count_kernel(TrackDims d, const int32_t* __restrict__ info, const int32_t* __restrict__ nkps,
                                                          int32_t* __restrict__ counts, int32_t* __restrict__ hist) {
    const int b = blockIdx.x, tid = threadIdx.x, s = max(0, b - d.n_kf + 1);
    __shared__ int red[4];
    __shared__ int h[kHist];
    if (tid < kHist) h[tid] = 0;
    __syncthreads();
    int nl = 0, ne = 0;
    for (int f = s; f <= b; ++f) {
        const int nkp = nkps ? min(max(nkps[f], 0), d.kp_cap) : d.kp_cap; // (slots beyond the frame's keypoints are never nodes)
        for (int i = tid; i < nkp; i += 256) {
            const int len = window_head_len(info[(size_t)f * d.kp_cap + i], s, b, f);
            nl += len > 0; ne += len;
            if (len > 0) atomicAdd(&h[min(len, kHist - 1)], 1); // (integer: order-free)
        }
    }
    nl = block_sum_i32(nl, red);
    ne = block_sum_i32(ne, red);
    if (tid == 0) { counts[2 * b] = nl; counts[2 * b + 1] = ne; }
    if (tid < kHist) hist[(size_t)b * kHist + tid] = h[tid];
}

// ---- offsets of the concatenated arrays (exclusive scan over the windows; one workgroup).  A window that would run past a capacity, and
// every window after it, is emitted EMPTY and the status word is set: the caller sized its arrays too small.
__global__ __launch_bounds__(256) void window_scan_kernel(TrackDims d, const int32_t* __restrict__ counts, int lm_capacity, int edge_capacity,
                                                         int32_t* __restrict__ lm_off, int32_t* __restrict__ edge_off, int32_t* __restrict__ n_kf_out,
                                                         int32_t* __restrict__ status) {
    __shared__ int sl[256], se[256];
    __shared__ int carry_l, carry_e, cut;
    const int tid = threadIdx.x;
    if (tid == 0) { carry_l = 0; carry_e = 0; cut = 0; lm_off[0] = 0; edge_off[0] = 0; }
    __syncthreads();
    for (int base = 0; base < d.B; base += 256) {
        const int b = base + tid;
        const int cl = b < d.B ? counts[2 * b] : 0, ce = b < d.B ? counts[2 * b + 1] : 0;
        sl[tid] = cl; se[tid] = ce;
        __syncthreads();
        for (int dd = 1; dd < 256; dd <<= 1) {
            const int a = tid >= dd ? sl[tid - dd] : 0, e = tid >= dd ? se[tid - dd] : 0;
            __syncthreads();
            sl[tid] += a; se[tid] += e;
            __syncthreads();
        }
        const int il = carry_l + sl[tid], ie = carry_e + se[tid]; // inclusive
        const bool over = il > lm_capacity || ie > edge_capacity;
        if (b < d.B && over) atomicExch(&cut, 1);
        __syncthreads();
        if (b < d.B) {
            // (prefixes are monotone: once a window overflows all later ones do; an overflowing window repeats the last good offset)
            int ol = il, oe = ie;
            if (over) { ol = -1; oe = -1; }
            lm_off[b + 1] = ol; edge_off[b + 1] = oe;
            n_kf_out[b] = min(b + 1, d.n_kf);
        }
        __syncthreads();
        if (tid == 255) { carry_l = il; carry_e = ie; }
        __syncthreads();
    }
    // second pass: replace the -1 marks by the last good offset (windows from the first overflow on are empty)
    __syncthreads();
    if (tid == 0) {
        if (cut) {
            int gl = 0, ge = 0;
            for (int b = 0; b < d.B; ++b) {
                if (lm_off[b + 1] < 0) { lm_off[b + 1] = gl; edge_off[b + 1] = ge; }
                else { gl = lm_off[b + 1]; ge = edge_off[b + 1]; }
            }
        }
        *status = cut ? 1 : 0;
    }
}

// ---- per window: poses, and the RANK of every chain head = its window-local landmark index (by observation count, then by (frame,
// keypoint)); the head's record goes to head_rec[global landmark index] for the emit pass.  No dependent loads here: one info word per
// slot, a ballot per count, three barriers per 1024 slots.
constexpr int kRankBlock = 1024, kRankWaves = kRankBlock / 64;
__global__ __launch_bounds__(kRankBlock) void window_rank_kernel(TrackDims d, const double* __restrict__ G, const int32_t* __restrict__ counts,
                                                                const int32_t* __restrict__ hist, const int32_t* __restrict__ info,
                                                                const int32_t* __restrict__ nkps, const int32_t* __restrict__ lm_off,
                                                                const int32_t* __restrict__ edge_off, double* __restrict__ T_out,
                                                                uint32_t* __restrict__ head_rec) {
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, s = max(0, b - d.n_kf + 1), nk = b - s + 1;
    __shared__ int s_c[kRankWaves][kHist];
    __shared__ int bin_l[kHist], s_run[kHist];
    for (int i = tid; i < d.n_kf * 7; i += kRankBlock) { // poses of the window's keyframes (unused slots: identity)
        const int k = i / 7, c = i - 7 * k;
        T_out[(size_t)b * d.n_kf * 7 + i] = k < nk ? G[(size_t)(s + k) * 7 + c] : (c == 3 ? 1.0 : 0.0);
    }
    const int l0 = lm_off[b];
    if (lm_off[b + 1] - l0 != counts[2 * b] || edge_off[b + 1] - edge_off[b] != counts[2 * b + 1]) return; // truncated by the capacity check: empty window
    if (tid == 0) { int al = 0; for (int c = 1; c < kHist; ++c) { bin_l[c] = al; al += hist[(size_t)b * kHist + c]; } }
    if (tid < kHist) s_run[tid] = 0;
    __syncthreads();
    for (int f = s; f <= b; ++f) {
        const int nkp = nkps ? min(max(nkps[f], 0), d.kp_cap) : d.kp_cap;
        for (int base = 0; base < nkp; base += kRankBlock) {
            const int i = base + tid;
            const int len = i < nkp ? min(window_head_len(info[(size_t)f * d.kp_cap + i], s, b, f), kHist - 1) : 0;
            int my_rank = 0, my_wave_cnt = 0;
#pragma unroll
            for (int c = 1; c < kHist; ++c) {
                const unsigned long long m = __ballot(len == c);
                if (len == c) my_rank = __popcll(m & ((1ull << lane) - 1ull));
                if (lane == c) my_wave_cnt = __popcll(m);
            }
            __syncthreads();
            if (lane < kHist) s_c[wave][lane] = my_wave_cnt;
            __syncthreads();
            if (len > 0) {
                int before = s_run[len];
                for (int w = 0; w < wave; ++w) before += s_c[w][len];
                head_rec[l0 + bin_l[len] + before + my_rank] = (uint32_t)(f - s) | ((uint32_t)i << 4) | ((uint32_t)len << 20);
            }
            __syncthreads();
            if (tid < kHist) { int t = 0; for (int w = 0; w < kRankWaves; ++w) t += s_c[w][tid]; s_run[tid] += t; }
        }
    }
}

// ---- one thread per landmark of the batch: its edges (chronological) and its position / reliable_depth_ as of its last observation inside
// the window.  Flat over the concatenated landmark array: the window is found by bisection of lm_off.
__global__ __launch_bounds__(256) void window_emit_kernel(TrackDims d, const vslam_keypoint* __restrict__ d_kps, const float* __restrict__ d_xyz,
                                                         const int32_t* __restrict__ kp2lr, const int32_t* __restrict__ root,
                                                         const int32_t* __restrict__ relsrc, const int32_t* __restrict__ succ,
                                                         const double* __restrict__ G, const float* __restrict__ carry, const int32_t* __restrict__ hist,
                                                         const uint32_t* __restrict__ head_rec, const int32_t* __restrict__ lm_off,
                                                         const int32_t* __restrict__ edge_off, float* __restrict__ xyz_out,
                                                         uint8_t* __restrict__ rel_out, uint8_t* __restrict__ inl_out, int32_t* __restrict__ kf_out,
                                                         int32_t* __restrict__ lm_out, float* __restrict__ uv_out) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= lm_off[d.B]) return;
    int lo = 0, hi = d.B; // largest b with lm_off[b] <= g
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (lm_off[mid] <= g) lo = mid; else hi = mid; }
    const int b = lo, s = max(0, b - d.n_kf + 1), l = g - lm_off[b];
    const uint32_t rec = head_rec[g];
    const int f = s + (int)(rec & 15u), i = (int)((rec >> 4) & 0xFFFFu), len = (int)(rec >> 20);
    int bl = 0, be = 0; // first landmark / first edge of the landmarks with `len` observations
    for (int c = 1; c < len; ++c) { const int n = hist[(size_t)b * kHist + c]; bl += n; be += n * c; }
    int e = edge_off[b] + be + (l - bl) * len;
    int cf = f, ci = i;
    for (int k = 0; k < len; ++k) {
        const vslam_keypoint* kp = d_kps + (size_t)cf * d.kp_cap + ci;
        kf_out[e] = cf - s; lm_out[e] = l;
        reinterpret_cast<float2*>(uv_out)[e] = make_float2(kp->x, kp->y);
        ++e;
        if (k + 1 < len) { ci = succ[(size_t)cf * d.kp_cap + ci]; ++cf; }
    }
    const size_t last = (size_t)cf * d.kp_cap + ci;
    const int rs = relsrc[last];
    const bool has_rel = rs >= 0 || rs <= kCarryCode;
    float pos[3];
    landmark_position(d, has_rel ? rs : root[last], kp2lr, d_xyz, G, carry, pos);
    float* o = xyz_out + 3 * (size_t)g;
    o[0] = pos[0]; o[1] = pos[1]; o[2] = pos[2];
    rel_out[g] = has_rel; inl_out[g] = 1;
}

size_t track_scratch_bytes(int B, int kp_cap, int lm_capacity) {
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    return al((size_t)lm_capacity * 4) + 6 * al((size_t)B * kp_cap * 4) + al((size_t)B * 7 * 8) + al((size_t)B * 2 * 4) + al((size_t)B * (VSLAM_MAX_KF + 1) * 4);
}

int launch_build_windows(const vslam_tracks_in& in, int n_kf, int lm_capacity, int edge_capacity, const double K4[4], double reproj_thr, int track_rule, uint8_t* scratch,
                         int32_t* d_lm_off, int32_t* d_edge_off, int32_t* d_n_kf, double* d_T, float* d_xyz_out, uint8_t* d_rel_out, uint8_t* d_inl_out,
                         int32_t* d_kf_out, int32_t* d_lm_out, float* d_uv_out, int32_t* d_status, hipStream_t stream) {
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    TrackDims d;
    d.B = in.n_frames; d.kp_cap = in.kp_capacity; d.lr_cap = in.lr_capacity; d.match_cap = in.match_capacity; d.pnp_cap = in.pnp_capacity; d.n_kf = n_kf;
    const size_t tab = al((size_t)d.B * d.kp_cap * 4);
    int32_t* kp2lr = (int32_t*)scratch; int32_t* pred = (int32_t*)(scratch + tab); int32_t* succ = (int32_t*)(scratch + 2 * tab);
    int32_t* root = (int32_t*)(scratch + 3 * tab); int32_t* relsrc = (int32_t*)(scratch + 4 * tab); int32_t* info = (int32_t*)(scratch + 5 * tab);
    double* G = (double*)(scratch + 6 * tab); int32_t* counts = (int32_t*)(scratch + 6 * tab + al((size_t)d.B * 7 * 8));
    int32_t* hist = (int32_t*)((uint8_t*)counts + al((size_t)d.B * 2 * 4));
    uint32_t* head_rec = (uint32_t*)((uint8_t*)hist + al((size_t)d.B * (VSLAM_MAX_KF + 1) * 4));
    TrackCam cam;
    cam.fx = K4[0]; cam.fy = K4[1]; cam.cx = K4[2]; cam.cy = K4[3]; cam.thr2 = reproj_thr * reproj_thr; cam.track_rule = track_rule;
    int32_t* cand = info; // (the candidate words live in the info table until track_info_kernel writes it)
    ProfScope prof__(stream, "build_windows_kernels", 9);
    hipLaunchKernelGGL(track_init_kernel, dim3(d.B), dim3(256), 0, stream, d, in.d_lr, in.d_nlr, kp2lr, pred, succ, cand);
    if (in.d_T_abs) VS_HIP(hipMemcpyAsync(G, in.d_T_abs, sizeof(double) * 7 * (size_t)d.B, hipMemcpyDeviceToDevice, stream)); // (a chunk: poses in the sequence's world)
    else hipLaunchKernelGGL(track_pose_chain_kernel, dim3(1), dim3(256), 0, stream, d.B, in.d_T_rel, G);
    if (d.B > 1) hipLaunchKernelGGL(track_link_kernel, dim3(d.B - 1), dim3(256), 0, stream, d, in.d_f2f, in.d_nf2f, in.d_valid, in.d_pose_inlier, kp2lr, cand, succ);
    hipLaunchKernelGGL(track_walk_kernel, dim3((d.kp_cap + 255) / 256, d.B), dim3(256), 0, stream, d, cam, in.d_kps, in.d_xyz, in.d_valid, in.d_reliable, kp2lr, cand, pred, succ, G,
                       in.d_carry_in, root, relsrc);
    hipLaunchKernelGGL(track_info_kernel, dim3((d.kp_cap + 255) / 256, d.B), dim3(256), 0, stream, d, pred, succ, root, in.d_carry_in, info);
    if (in.d_carry_out && in.carry_out_frame > 0 && in.carry_out_frame < d.B)
        hipLaunchKernelGGL(track_carry_out_kernel, dim3((d.kp_cap + 255) / 256), dim3(256), 0, stream, d, in.carry_out_frame, kp2lr, pred, root, relsrc, in.d_xyz, G,
                           in.d_carry_in, in.d_carry_out);
    hipLaunchKernelGGL(window_count_kernel, dim3(d.B), dim3(256), 0, stream, d, info, in.d_nkps, counts, hist);
    hipLaunchKernelGGL(window_scan_kernel, dim3(1), dim3(256), 0, stream, d, counts, lm_capacity, edge_capacity, d_lm_off, d_edge_off, d_n_kf, d_status);
    hipLaunchKernelGGL(window_rank_kernel, dim3(d.B), dim3(kRankBlock), 0, stream, d, G, counts, hist, info, in.d_nkps, d_lm_off, d_edge_off, d_T, head_rec);
    hipLaunchKernelGGL(window_emit_kernel, dim3((lm_capacity + 255) / 256), dim3(256), 0, stream, d, in.d_kps, in.d_xyz, kp2lr, root, relsrc, succ, G, in.d_carry_in, hist, head_rec,
                       d_lm_off, d_edge_off, d_xyz_out, d_rel_out, d_inl_out, d_kf_out, d_lm_out, d_uv_out);
    VS_HIP(hipGetLastError());
    return VSLAM_OK;
}

} // namespace vslam
