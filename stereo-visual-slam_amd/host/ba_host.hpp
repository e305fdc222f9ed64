// ba_host.hpp -- C++ mirror of optimization.hpp's two entry points
// (/root/reference/include/stereo_visual_slam_main/optimization.hpp:137-152) on top of the C-ABI.
#pragma once
#include <array>
#include <unordered_map>

#include "types.hpp"

namespace vslam {

// K: the reference's `const cv::Mat& K` (3x3, run_vslam.cpp:34-38) as a row-major 3x3 array; fx, fy, cx, cy are read from it
// and handed to the C-ABI per call, so a caller with other intrinsics than the context's is honoured.  q1_quirk reproduces the reference's use of Feature::feature_id_ as a vector
// index (optimization.cpp:170); false looks the feature up by id instead.
using Mat33 = std::array<double, 9>; // stands in for the CV_64F 3x3 cv::Mat K
void optimize_map(vslam_ctx* ctx, std::unordered_map<unsigned long, Frame>& keyframes, std::unordered_map<unsigned long, Landmark>& landmarks,
                  const Mat33& K, bool if_update_map, bool if_update_landmark, int num_ite, bool q1_quirk = true);
void optimize_pose_only(vslam_ctx* ctx, std::unordered_map<unsigned long, Frame>& keyframes, std::unordered_map<unsigned long, Landmark>& landmarks,
                        const Mat33& K, bool if_update_map, int num_ite, bool q1_quirk = true);


// The reference's exact signatures (optimization.hpp:137-139, :150-152): no context, no quirk switch.  They run on the process-level
// backend below, which VO's constructor sets to its own context (the reference's optimisers are free functions with no state either;
// the GPU context is this build's only addition and lives behind this accessor).
void set_optimizer_backend(vslam_ctx* ctx, bool q1_quirk = true);
void clear_optimizer_backend(vslam_ctx* ctx); // no-op unless ctx is the bound one; call before vslam_destroy(ctx)
vslam_ctx* optimizer_backend();
void optimize_map(std::unordered_map<unsigned long, Frame>& keyframes, std::unordered_map<unsigned long, Landmark>& landmarks, const Mat33& K,
                  bool if_update_map, bool if_update_landmark, int num_ite);
void optimize_pose_only(std::unordered_map<unsigned long, Frame>& keyframes, std::unordered_map<unsigned long, Landmark>& landmarks, const Mat33& K,
                        bool if_update_map, int num_ite);

} // namespace vslam
