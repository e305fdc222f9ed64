// map_host.hpp -- C++ mirror of vslam::Map (/root/reference/include/stereo_visual_slam_main/map.hpp:15-81) without ROS.
// Same public members and methods; the rviz publishers are dropped (out of scope), the trajectory writer is kept.
#pragma once
#include <string>
#include <unordered_map>

#include "types.hpp"

namespace vslam {

struct Map {
    std::unordered_map<unsigned long, Frame> keyframes_;
    std::unordered_map<unsigned long, Landmark> landmarks_;
    const int num_keyframes_ = 10;   // map.hpp:22
    int current_keyframe_id_ = 0;
    bool if_write_pose_ = false;
    std::string traj_path_ = "estimated_traj.txt";

    explicit Map(bool if_write_pose = false, std::string traj_path = "estimated_traj.txt") : if_write_pose_(if_write_pose), traj_path_(std::move(traj_path)) {}

    int insert_keyframe(Frame frame_to_add);
    int insert_landmark(Landmark landmark_to_add);
    int remove_keyframe();
    int clean_map();
    void write_pose(const Frame& frame);
    void write_remaining_pose();
};

} // namespace vslam
