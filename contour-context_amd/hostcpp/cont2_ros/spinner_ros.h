// Stand-in for include/cont2_ros/spinner_ros.h of the reference so that its offline driver (test/batch_bin_test.cpp,
// built with PUB_ROS_MSG=0) compiles and runs against this mirror without ROS: the few `ros::` names the driver
// touches (init, NodeHandle, Time, Rate, Duration, ok, spinOnce) and the non-ROS members of BaseROSSpinner it reads
// (pause / terminate flags, the pose bookkeeping map).  No topics, no rviz: those are outside the hot path.
//
// One behavioural choice: the reference loop runs `while (ros::ok())` and sleeps with ros::Duration(1.0).sleep() when the
// scan list is exhausted (spinOnce() == 1), waiting for a "terminate" message on a topic.  Without a ROS master nobody
// can send it, so here that sleep ends the loop: Duration::sleep() makes ros::ok() false.
#pragma once
#include <chrono>
#include <cstdint>
#include <map>
#include <mutex>
#include <string>
#include <thread>

#include "../compat/compat.h"

namespace ros {
namespace detail {
inline bool &ok_flag() {
  static bool ok = true;
  return ok;
}
}  // namespace detail
inline void init(int &, char **, const std::string &) {}
inline bool ok() { return detail::ok_flag(); }
inline void spinOnce() {}
inline void shutdown() { detail::ok_flag() = false; }
struct NodeHandle {};
struct Time {
  double sec = 0;
  static Time now() {
    Time t;
    t.sec = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    return t;
  }
};
struct Duration {
  double s;
  explicit Duration(double s_) : s(s_) {}
  void sleep() const { detail::ok_flag() = false; }  // see the header comment
};
struct Rate {
  explicit Rate(double) {}
  void sleep() const {}  // an offline replay has nobody to wait for
};
}  // namespace ros

struct BaseROSSpinner {
  struct GlobalPoseInfo {
    Eigen::Isometry3d T_wl;
    double z_shift{};
    GlobalPoseInfo(const Eigen::Isometry3d &a, const double &b) : T_wl(a), z_shift(b) {}
    GlobalPoseInfo() = default;
  };
  ros::NodeHandle nh;
  std::map<int, GlobalPoseInfo> g_poses;
  uint64_t lc_line_cnt = 0;
  bool stat_paused = false;
  bool stat_terminated = false;
  std::mutex mtx_status;
  explicit BaseROSSpinner(ros::NodeHandle &nh_) : nh(nh_) {}
};
