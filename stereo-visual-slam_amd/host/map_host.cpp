// map_host.cpp -- keyframe window bookkeeping, semantics of /root/reference/src/stereo_visual_slam_main/map.cpp:13-204.
// Pointer-chasing O(10) host work: stays on the CPU by design (SURVEY.md section 2 "Map management").
#include "map_host.hpp"

#include <algorithm>
#include <fstream>
#include <limits>

namespace vslam {

int Map::insert_keyframe(Frame frame_to_add) {
    current_keyframe_id_ = frame_to_add.keyframe_id_;
    keyframes_[(unsigned long)frame_to_add.keyframe_id_] = std::move(frame_to_add); // insert or overwrite (map.cpp:17-24)
    if ((int)keyframes_.size() > num_keyframes_) remove_keyframe();                  // map.cpp:27-30
    return 0;
}

int Map::insert_landmark(Landmark landmark_to_add) {
    landmarks_[(unsigned long)landmark_to_add.landmark_id_] = std::move(landmark_to_add);
    return 0;
}

int Map::remove_keyframe() {
    // the keyframe nearest to the current one if closer than 0.2 (tangent norm), otherwise the farthest (map.cpp:50-86)
    double far_d = 0, near_d = 1000000;
    unsigned long far_id = 0, near_id = 0;
    const SE3 T_w_c = keyframes_.at(current_keyframe_id_).T_c_w_.inverse();
    for (const auto& kv : keyframes_) {
        if ((int)kv.first == current_keyframe_id_) continue;
        const double d = norm6((kv.second.T_c_w_ * T_w_c).log());
        if (d > far_d) { far_d = d; far_id = kv.first; }
        if (d < near_d) { near_d = d; near_id = kv.first; }
    }
    const unsigned long victim = near_d < 0.2 ? near_id : far_id;
    // drop the victim's observations from the landmarks it sees (map.cpp:91-108)
    for (const Feature& f : keyframes_.at(victim).features_) {
        auto it = landmarks_.find((unsigned long)f.landmark_id_);
        if (it == landmarks_.end()) continue; // the reference would throw from .at(); a missing landmark is simply skipped
        auto& obs = it->second.observations_;
        obs.erase(std::remove_if(obs.begin(), obs.end(), [&](const Observation& o) { return o.keyframe_id_ == (int)victim && o.feature_id_ == f.feature_id_; }), obs.end());
        it->second.observed_times_--;
    }
    if (if_write_pose_) write_pose(keyframes_.at(victim)); // map.cpp:120-123
    keyframes_.erase(victim);
    clean_map();
    return 0;
}

int Map::clean_map() {
    for (auto it = landmarks_.begin(); it != landmarks_.end();) {
        if (it->second.observed_times_ == 0) it = landmarks_.erase(it); // map.cpp:138-142
        else ++it;
    }
    return 0;
}

void Map::write_pose(const Frame& frame) {
    // "frame_id r00 r01 r02 x r10 r11 r12 y r20 r21 r22 z" of T_w_c, appended (map.cpp:168-196)
    const SE3 T_w_c = frame.T_c_w_.inverse();
    const auto R = T_w_c.rotationMatrix();
    const auto t = T_w_c.translation();
    std::ofstream file(traj_path_, std::ios_base::app);
    file << frame.frame_id_ << " " << R[0] << " " << R[1] << " " << R[2] << " " << t[0] << " " << R[3] << " " << R[4] << " " << R[5] << " " << t[1] << " "
         << R[6] << " " << R[7] << " " << R[8] << " " << t[2] << std::endl;
}

void Map::write_remaining_pose() {
    for (const auto& kv : keyframes_) write_pose(kv.second);
}

} // namespace vslam
