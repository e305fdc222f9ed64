// ba_host.hpp -- C++ mirror of optimization.hpp's two entry points
// (/root/reference/include/stereo_visual_slam_main/optimization.hpp:137-152) on top of the C-ABI.
#pragma once
#include <unordered_map>

#include "types.hpp"

namespace vslam {

// K: {fx, fy, cx, cy} is taken from the context parameters (the reference passes a 3x3 cv::Mat that always holds the
// KITTI constants, run_vslam.cpp:34-38).  q1_quirk reproduces the reference's use of Feature::feature_id_ as a vector
// index (optimization.cpp:170); false looks the feature up by id instead.
void optimize_map(vslam_ctx* ctx, std::unordered_map<unsigned long, Frame>& keyframes, std::unordered_map<unsigned long, Landmark>& landmarks,
                  bool if_update_map, bool if_update_landmark, int num_ite, bool q1_quirk = true);
void optimize_pose_only(vslam_ctx* ctx, std::unordered_map<unsigned long, Frame>& keyframes, std::unordered_map<unsigned long, Landmark>& landmarks,
                        bool if_update_map, int num_ite, bool q1_quirk = true);

} // namespace vslam
