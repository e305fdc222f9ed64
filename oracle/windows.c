/* windows.c -- CPU restatement of the map bookkeeping that feeds the local BA, for a batch of consecutive keyframes.
 * TEST INFRASTRUCTURE ONLY (see vo_oracle.h).  PARITY UNPINNED like the rest of the oracle; what is restated here is the REFERENCE'S OWN
 * code, followed statement by statement:
 *   VO::tracking            visual_odometry.cpp:592-599  a frame-to-frame match gives the current feature the landmark of the matched
 *                                                         feature of the last frame
 *   VO::tracking            :568-574                      the last frame's query set is EVERY feature of the last frame -- tracked or created there -- whether
 *                                                         or not it has a depth of its own in that frame (track_rule 1, round 6; see below)
 *   VO::motion_estimation   :260-270                      the 3-D point of a tracked feature is its LANDMARK's map position (pt_3d_)
 *   VO::motion_estimation   :306                          outliers of the pose stage are erased from the frame
 *   VO::insert_key_frame    :363-372                      every remaining feature adds an Observation to its landmark
 *                           :381-421                      every keypoint with a valid depth that is not such a feature creates a Landmark
 *                                                         (position = T_c_w^-1 * p_c, reliable_depth_); an existing landmark whose depth was
 *                                                         unreliable takes the first reliable one (:391-401)
 *   optimize_map            optimization.cpp:127-214      vertices = keyframes of the map, landmarks with their observations, one edge per
 *                                                         observation (the is_inlier / reliable_depth_ filter of :160 is applied by the
 *                                                         optimiser on the flags emitted here)
 * Deliberately written the way the reference is -- one pass over the frames in time order that maintains landmark records with
 * observation lists -- NOT the way the HIP path is (flat predecessor / successor tables, one thread per keypoint slot, per-window ordered
 * compaction): the two decompositions share nothing but the rule set.
 * Throughput-mode conventions shared with the HIP path: every frame is a keyframe; window b = keyframes [max(0, b - n_kf + 1), b] with
 * the map state right after keyframe b; is_inlier = 1 on entry; world = frame 0, poses = the pose stage's relative poses chained
 * sequentially; landmarks of a window ordered by the number of observations inside it, then by their first observation inside it (frame, then
 * keypoint index) -- the optimiser takes any order (the reference iterates an unordered_map), this one keeps neighbouring landmarks alike.
 *
 * Track continuation (round 6).  track_rule 0 is the convention of rounds 4-5: a frame-to-frame match continues a track only when the LAST-frame
 * keypoint owns a valid depth of its own (it then is one of the pose stage's inputs, and the pose stage's inlier flag decides).  track_rule 1 is the
 * reference's bookkeeping: the match continues a track whenever the last-frame keypoint IS A FEATURE (carries a landmark), created in that frame or
 * tracked into it.  For a feature without a depth of its own the pose stage of throughput mode has no input (its inputs are triangulated in the
 * last frame's camera, all frame pairs of a batch at once); the reference would hand solvePnPRansac the landmark's map position (:268), so the same
 * inlier rule (:277, reprojection error <= 4 px) is applied to exactly that: the landmark's position as the map holds it before this frame's
 * insertion (pt_3d_: the creation point, or the first reliable one, :391-401), projected with the frame's chained pose.  The pose itself is not
 * re-estimated.  The pose stage's own inputs and their flags are unchanged. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "vo_oracle.h"

typedef struct { int frame, kp; } obs_t;
typedef struct {
    int root_frame;
    float pos0[3], pos1[3]; /* position at creation / after the reliable-depth update */
    int rel_frame;          /* frame from which reliable_depth_ is true (root_frame if created reliable), -1: never */
    int n_obs, cap_obs;
    obs_t* obs;
} lmk_t;

static void push_obs(lmk_t* L, int f, int kp) {
    if (L->n_obs == L->cap_obs) { L->cap_obs = L->cap_obs ? 2 * L->cap_obs : 4; L->obs = (obs_t*)realloc(L->obs, sizeof(obs_t) * (size_t)L->cap_obs); }
    L->obs[L->n_obs].frame = f; L->obs[L->n_obs].kp = kp; ++L->n_obs;
}

static void world_point(const double* G /* T_c_w of the frame */, const float* pc, float out[3]) {
    double Gi[7], p[3] = {pc[0], pc[1], pc[2]}, pw[3];
    vo_se3_inv(G, Gi);
    vo_se3_act(Gi, p, pw);
    out[0] = (float)pw[0]; out[1] = (float)pw[1]; out[2] = (float)pw[2];
}

typedef struct { int cnt, first_frame, first_kp, id; } head_t;
static int head_cmp(const void* a, const void* b) {
    const head_t* x = (const head_t*)a; const head_t* y = (const head_t*)b;
    if (x->cnt != y->cnt) return x->cnt < y->cnt ? -1 : 1;
    if (x->first_frame != y->first_frame) return x->first_frame < y->first_frame ? -1 : 1;
    return x->first_kp < y->first_kp ? -1 : (x->first_kp > y->first_kp);
}

/* PnPRansac's inlier test (visual_odometry.cpp:277: reprojection error 4.0) on a landmark's map position (cv::Point3f) seen through T_c_w */
static int reprojects_within(const float pos[3], const double* T_c_w, const vo_keypoint* kp, const double K4[4], double thr) {
    const double pw[3] = {pos[0], pos[1], pos[2]};
    double pc[3];
    vo_se3_act(T_c_w, pw, pc);
    const double du = (double)kp->x - (K4[0] * pc[0] / pc[2] + K4[2]), dv = (double)kp->y - (K4[1] * pc[1] / pc[2] + K4[3]);
    const double c = du * du + dv * dv;
    return isfinite(c) && c <= thr * thr;
}

int vo_build_windows(int n_frames, int kp_cap, int lr_cap, int match_cap, int pnp_cap, const vo_keypoint* kps, const vo_dmatch* lr,
                     const int32_t* nlr, const float* xyz, const uint8_t* valid, const uint8_t* reliable, const vo_dmatch* f2f,
                     const int32_t* nf2f, const uint8_t* pose_inlier, const double* T_rel, int n_kf, int lm_capacity, int edge_capacity,
                     int32_t* lm_off, int32_t* edge_off, int32_t* n_kf_out, double* T_out, float* xyz_out, uint8_t* rel_out,
                     uint8_t* inl_out, int32_t* kf_out, int32_t* lm_out, float* uv_out, const double K4[4], double reproj_thr, int track_rule) {
    if (n_frames <= 0 || n_kf <= 0 || (track_rule && !K4)) return -1;
    double* G = (double*)malloc(sizeof(double) * 7 * (size_t)n_frames);
    int32_t* feat_lm = (int32_t*)malloc(sizeof(int32_t) * (size_t)n_frames * kp_cap); /* landmark of keypoint (f, i), -1: not a feature */
    int32_t* kp2lr = (int32_t*)malloc(sizeof(int32_t) * (size_t)kp_cap * 2);
    int n_lm = 0, cap_lm = 1024;
    lmk_t* L = (lmk_t*)calloc((size_t)cap_lm, sizeof(lmk_t));
    for (size_t i = 0; i < (size_t)n_frames * kp_cap; ++i) feat_lm[i] = -1;
    int32_t* prev_k2 = kp2lr; int32_t* cur_k2 = kp2lr + kp_cap;
    for (int f = 0; f < n_frames; ++f) {
        /* pose: T_c_w of frame f = T_{f,f-1} * T_c_w of frame f - 1 (tracking: frame_current_.T_c_w_ = T_c_w_, :612) */
        if (f == 0) { G[0] = G[1] = G[2] = 0; G[3] = 1; G[4] = G[5] = G[6] = 0; }
        else vo_se3_mul(T_rel + 7 * (size_t)(f - 1), G + 7 * (size_t)(f - 1), G + 7 * (size_t)f);
        { int32_t* t = prev_k2; prev_k2 = cur_k2; cur_k2 = t; }
        for (int i = 0; i < kp_cap; ++i) cur_k2[i] = -1;
        const int n = nlr[f] < 0 ? 0 : (nlr[f] > lr_cap ? lr_cap : nlr[f]);
        for (int m = 0; m < n; ++m) { const int q = lr[(size_t)f * lr_cap + m].queryIdx; if (q >= 0 && q < kp_cap) cur_k2[q] = m; }
        /* tracked features: the pose stage's inputs are the matches whose last-frame keypoint owns a valid depth, in match order */
        if (f > 0) {
            const int it = f - 1;
            const int nm = nf2f[it] < 0 ? 0 : (nf2f[it] > match_cap ? match_cap : nf2f[it]);
            int j = 0;
            for (int k = 0; k < nm; ++k) {
                const int q = f2f[(size_t)it * match_cap + k].queryIdx, t = f2f[(size_t)it * match_cap + k].trainIdx;
                if (q < 0 || q >= kp_cap || t < 0 || t >= kp_cap) continue;
                const int li = prev_k2[q];
                const int id = feat_lm[(size_t)it * kp_cap + q];
                if (li >= 0 && valid[(size_t)it * lr_cap + li]) { /* an input of the pose stage: its flag decides */
                    const int jj = j++;
                    if (jj >= pnp_cap || !pose_inlier[(size_t)it * pnp_cap + jj]) continue;
                    /* a keypoint with a valid depth always is a feature of its keyframe */
                    if (id < 0) { free(G); free(feat_lm); free(kp2lr); for (int x = 0; x < n_lm; ++x) free(L[x].obs); free(L); return -2; }
                } else { /* no depth of its own in the last frame: a feature only if it was tracked into it (:568-574 takes every feature) */
                    if (!track_rule || id < 0) continue;
                    const float* pos = (L[id].rel_frame >= 0 && L[id].rel_frame != L[id].root_frame) ? L[id].pos1 : L[id].pos0; /* pt_3d_ before this frame's insertion */
                    if (!reprojects_within(pos, G + 7 * (size_t)f, &kps[(size_t)f * kp_cap + t], K4, reproj_thr)) continue;
                }
                feat_lm[(size_t)f * kp_cap + t] = id;
                push_obs(&L[id], f, t); /* :363-372 */
            }
        }
        /* keypoints with a valid depth: update an existing landmark's unreliable depth, or create one (:381-421) */
        for (int i = 0; i < kp_cap; ++i) {
            const int m = cur_k2[i];
            if (m < 0 || !valid[(size_t)f * lr_cap + m]) continue;
            const int rel = reliable[(size_t)f * lr_cap + m] != 0;
            const int id = feat_lm[(size_t)f * kp_cap + i];
            if (id >= 0) {
                if (L[id].rel_frame < 0 && rel) { world_point(G + 7 * (size_t)f, xyz + 3 * ((size_t)f * lr_cap + m), L[id].pos1); L[id].rel_frame = f; }
                continue;
            }
            if (n_lm == cap_lm) { cap_lm *= 2; L = (lmk_t*)realloc(L, sizeof(lmk_t) * (size_t)cap_lm); memset(L + n_lm, 0, sizeof(lmk_t) * (size_t)(cap_lm - n_lm)); }
            lmk_t* nl = &L[n_lm];
            nl->root_frame = f; nl->rel_frame = rel ? f : -1; nl->n_obs = 0; nl->cap_obs = 0; nl->obs = NULL;
            world_point(G + 7 * (size_t)f, xyz + 3 * ((size_t)f * lr_cap + m), nl->pos0);
            memcpy(nl->pos1, nl->pos0, sizeof(nl->pos0));
            push_obs(nl, f, i);
            feat_lm[(size_t)f * kp_cap + i] = n_lm++;
        }
    }
    /* windows: the graph optimize_map would build from the map after keyframe b */
    head_t* heads = (head_t*)malloc(sizeof(head_t) * (size_t)(n_lm > 0 ? n_lm : 1));
    int status = 0, tot_l = 0, tot_e = 0;
    lm_off[0] = 0; edge_off[0] = 0;
    for (int b = 0; b < n_frames; ++b) {
        const int s = b - n_kf + 1 < 0 ? 0 : b - n_kf + 1, nk = b - s + 1;
        n_kf_out[b] = nk;
        for (int k = 0; k < n_kf; ++k)
            for (int c = 0; c < 7; ++c) T_out[((size_t)b * n_kf + k) * 7 + c] = k < nk ? G[7 * (size_t)(s + k) + c] : (c == 3 ? 1.0 : 0.0);
        int nh = 0, ne = 0;
        for (int id = 0; id < n_lm && !status; ++id) {
            if (L[id].root_frame > b) break; /* landmarks are created in frame order */
            int first = -1, cnt = 0;
            for (int o = 0; o < L[id].n_obs; ++o)
                if (L[id].obs[o].frame >= s && L[id].obs[o].frame <= b) { if (first < 0) first = o; ++cnt; }
            if (first < 0) continue;
            heads[nh].first_frame = L[id].obs[first].frame; heads[nh].first_kp = L[id].obs[first].kp; heads[nh].id = id; heads[nh].cnt = cnt; ++nh;
            ne += cnt;
        }
        if (status || tot_l + nh > lm_capacity || tot_e + ne > edge_capacity) { status = 1; lm_off[b + 1] = tot_l; edge_off[b + 1] = tot_e; continue; }
        qsort(heads, (size_t)nh, sizeof(head_t), head_cmp);
        int e = tot_e;
        for (int l = 0; l < nh; ++l) {
            const lmk_t* lk = &L[heads[l].id];
            const int rel_now = lk->rel_frame >= 0 && lk->rel_frame <= b;
            const float* pos = (rel_now && lk->rel_frame != lk->root_frame) ? lk->pos1 : lk->pos0;
            xyz_out[3 * (size_t)(tot_l + l)] = pos[0]; xyz_out[3 * (size_t)(tot_l + l) + 1] = pos[1]; xyz_out[3 * (size_t)(tot_l + l) + 2] = pos[2];
            rel_out[tot_l + l] = (uint8_t)rel_now; inl_out[tot_l + l] = 1;
            for (int o = 0; o < lk->n_obs; ++o) {
                const int of = lk->obs[o].frame;
                if (of < s || of > b) continue;
                kf_out[e] = of - s; lm_out[e] = l;
                uv_out[2 * (size_t)e] = kps[(size_t)of * kp_cap + lk->obs[o].kp].x; uv_out[2 * (size_t)e + 1] = kps[(size_t)of * kp_cap + lk->obs[o].kp].y;
                ++e;
            }
        }
        tot_l += nh; tot_e = e;
        lm_off[b + 1] = tot_l; edge_off[b + 1] = tot_e;
    }
    free(heads); free(G); free(feat_lm); free(kp2lr);
    for (int x = 0; x < n_lm; ++x) free(L[x].obs);
    free(L);
    return status;
}
