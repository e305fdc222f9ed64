// ba_host.cpp -- graph construction + write-back around vslam_local_ba / vslam_pose_only_window.
// Mirrors the host half of optimize_map / optimize_pose_only
// (/root/reference/src/stereo_visual_slam_main/optimization.cpp:127-214 graph build, :254-287 write-back); the
// optimiser itself (g2o in the reference) runs on the GPU.
#include "ba_host.hpp"

#include <algorithm>
#include <map>
#include <stdexcept>

namespace vslam {

namespace {

struct Graph {
    std::vector<unsigned long> kf_ids, lm_ids;      // row -> map key
    std::vector<double> T;                          // n_kf x 7
    std::vector<float> xyz, uv;                     // n_lm x 3, n_edge x 2
    std::vector<int32_t> kf_idx, lm_idx, flag_lm;   // per edge
};

// pose_only: landmark filter is is_inlier only (optimization.cpp:334); otherwise is_inlier && reliable_depth_ (:160)
Graph build(std::unordered_map<unsigned long, Frame>& keyframes, std::unordered_map<unsigned long, Landmark>& landmarks, bool pose_only, bool q1_quirk) {
    Graph g;
    std::map<unsigned long, int> kf_row, lm_row; // ordered: deterministic row numbering (g2o orders vertices by id)
    for (const auto& kv : keyframes) kf_row[kv.first] = 0;
    for (auto& kv : kf_row) {
        kv.second = (int)g.kf_ids.size();
        g.kf_ids.push_back(kv.first);
        const double* d = keyframes.at(kv.first).T_c_w_.data();
        g.T.insert(g.T.end(), d, d + 7);
    }
    std::vector<unsigned long> lm_sorted;
    for (const auto& kv : landmarks) lm_sorted.push_back(kv.first);
    std::sort(lm_sorted.begin(), lm_sorted.end());
    // flag targets may be landmarks that are not in the graph: give every landmark touched a row in a side table
    std::map<unsigned long, int> flag_row;
    for (unsigned long id : lm_sorted) {
        const Landmark& lm = landmarks.at(id);
        if (!lm.is_inlier || (!pose_only && !lm.reliable_depth_)) continue;
        for (const Observation& obs : lm.observations_) {
            auto kf_it = keyframes.find((unsigned long)obs.keyframe_id_);
            if (kf_it == keyframes.end()) continue; // reference: std::out_of_range
            const std::vector<Feature>& feats = kf_it->second.features_;
            const Feature* feat = nullptr;
            if (q1_quirk) { // features_.at(obs.feature_id_): feature_id_ used as an index (optimization.cpp:170)
                if (obs.feature_id_ < 0 || obs.feature_id_ >= (int)feats.size()) continue;
                feat = &feats[(size_t)obs.feature_id_];
            } else {
                for (const Feature& f : feats) if (f.feature_id_ == obs.feature_id_ && f.landmark_id_ == (int)id) { feat = &f; break; }
                if (!feat) continue;
            }
            if (!lm_row.count(id)) {
                lm_row[id] = (int)g.lm_ids.size();
                g.lm_ids.push_back(id);
                g.xyz.push_back(lm.pt_3d_.x); g.xyz.push_back(lm.pt_3d_.y); g.xyz.push_back(lm.pt_3d_.z);
            }
            g.kf_idx.push_back(kf_row.at((unsigned long)obs.keyframe_id_));
            g.lm_idx.push_back(lm_row.at(id));
            g.uv.push_back(feat->keypoint_.pt.x); g.uv.push_back(feat->keypoint_.pt.y);
            g.flag_lm.push_back(feat->landmark_id_); // resolved to rows below (optimization.cpp:258-264)
        }
    }
    // flags may point at landmarks outside the graph (quirk Q1): append them as edge-less rows so they can be written
    for (int32_t& f : g.flag_lm) {
        const unsigned long id = (unsigned long)f;
        if (f < 0 || !landmarks.count(id)) { f = -1; continue; }
        if (!lm_row.count(id)) {
            lm_row[id] = (int)g.lm_ids.size();
            g.lm_ids.push_back(id);
            const Landmark& lm = landmarks.at(id);
            g.xyz.push_back(lm.pt_3d_.x); g.xyz.push_back(lm.pt_3d_.y); g.xyz.push_back(lm.pt_3d_.z);
        }
        f = lm_row.at(id);
    }
    return g;
}

void run(vslam_ctx* ctx, std::unordered_map<unsigned long, Frame>& keyframes, std::unordered_map<unsigned long, Landmark>& landmarks, const Mat33& K, bool pose_only,
         bool if_update_map, bool if_update_landmark, int num_ite, bool q1_quirk) {
    Graph g = build(keyframes, landmarks, pose_only, q1_quirk);
    if (g.kf_idx.empty() || g.kf_ids.empty()) return; // nothing to optimise (g2o would report "0 vertices")
    const int n_kf = (int)g.kf_ids.size(), n_lm = (int)g.lm_ids.size(), n_edge = (int)g.kf_idx.size();
    std::vector<uint8_t> inl((size_t)n_lm, 2); // 2 = untouched
    const double K4[4] = {K[0], K[4], K[2], K[5]}; // fx, fy, cx, cy out of the 3x3 (optimization.cpp:47-48 reads the same four entries)
    int rc;
    if (pose_only)
        rc = vslam_pose_only_window(ctx, n_kf, g.T.data(), n_lm, g.xyz.data(), n_edge, g.kf_idx.data(), g.lm_idx.data(), g.uv.data(), K4, g.flag_lm.data(), num_ite,
                                    if_update_map ? 1 : 0, inl.data(), nullptr, nullptr, nullptr);
    else
        rc = vslam_local_ba(ctx, n_kf, g.T.data(), n_lm, g.xyz.data(), n_edge, g.kf_idx.data(), g.lm_idx.data(), g.uv.data(), K4, g.flag_lm.data(), num_ite,
                            if_update_map ? 1 : 0, (if_update_map && if_update_landmark) ? 1 : 0, inl.data(), nullptr, nullptr, nullptr);
    if (rc != VSLAM_OK) throw std::runtime_error(std::string("window optimisation failed: ") + vslam_last_error());
    for (int l = 0; l < n_lm; ++l)
        if (inl[(size_t)l] != 2) landmarks.at(g.lm_ids[(size_t)l]).is_inlier = inl[(size_t)l] != 0;       // optimization.cpp:254-266
    if (if_update_map) {
        for (int k = 0; k < n_kf; ++k) keyframes.at(g.kf_ids[(size_t)k]).T_c_w_ = SE3(&g.T[(size_t)k * 7]); // :275-278
        if (!pose_only && if_update_landmark)
            for (int l = 0; l < n_lm; ++l) {
                Landmark& lm = landmarks.at(g.lm_ids[(size_t)l]);
                lm.pt_3d_ = Point3f(g.xyz[3 * (size_t)l], g.xyz[3 * (size_t)l + 1], g.xyz[3 * (size_t)l + 2]);   // :281-285
            }
    }
}

} // namespace

void optimize_map(vslam_ctx* ctx, std::unordered_map<unsigned long, Frame>& keyframes, std::unordered_map<unsigned long, Landmark>& landmarks,
                  const Mat33& K, bool if_update_map, bool if_update_landmark, int num_ite, bool q1_quirk) {
    run(ctx, keyframes, landmarks, K, false, if_update_map, if_update_landmark, num_ite, q1_quirk);
}

void optimize_pose_only(vslam_ctx* ctx, std::unordered_map<unsigned long, Frame>& keyframes, std::unordered_map<unsigned long, Landmark>& landmarks,
                        const Mat33& K, bool if_update_map, int num_ite, bool q1_quirk) {
    run(ctx, keyframes, landmarks, K, true, if_update_map, false, num_ite, q1_quirk);
}

namespace {
vslam_ctx* g_backend_ctx = nullptr;
bool g_backend_q1 = true;
} // namespace

void set_optimizer_backend(vslam_ctx* ctx, bool q1_quirk) { g_backend_ctx = ctx; g_backend_q1 = q1_quirk; }
void clear_optimizer_backend(vslam_ctx* ctx) { if (g_backend_ctx == ctx) g_backend_ctx = nullptr; }
vslam_ctx* optimizer_backend() { return g_backend_ctx; }

void optimize_map(std::unordered_map<unsigned long, Frame>& keyframes, std::unordered_map<unsigned long, Landmark>& landmarks, const Mat33& K,
                  bool if_update_map, bool if_update_landmark, int num_ite) {
    if (!g_backend_ctx) throw std::runtime_error("optimize_map: no optimiser backend (construct a VO or call set_optimizer_backend first)");
    run(g_backend_ctx, keyframes, landmarks, K, false, if_update_map, if_update_landmark, num_ite, g_backend_q1);
}

void optimize_pose_only(std::unordered_map<unsigned long, Frame>& keyframes, std::unordered_map<unsigned long, Landmark>& landmarks, const Mat33& K,
                        bool if_update_map, int num_ite) {
    if (!g_backend_ctx) throw std::runtime_error("optimize_pose_only: no optimiser backend (construct a VO or call set_optimizer_backend first)");
    run(g_backend_ctx, keyframes, landmarks, K, true, if_update_map, false, num_ite, g_backend_q1);
}

} // namespace vslam
