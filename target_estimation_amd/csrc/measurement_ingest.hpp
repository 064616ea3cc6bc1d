// measurement_ingest.hpp -- transport-agnostic restatement of the reference's ROS ingest policy:
// the per-id measurement mailbox (class Measurement, include/target_estimation/
// target_manager_ros.hpp:74-134), the /tf callback's id parsing (src/target_manager_ros.cpp:26-39,
// utils.hpp:273-313) and the per-tick policy of RosTargetManager::update
// (src/target_manager_ros.cpp:41-92): new measurement -> create if missing + predict/update;
// otherwise predict only; measurement timeout -> erase.  ROS itself (subscription, TF broadcast)
// is out of scope; the caller pushes (frame name | id, stamp, pose) and gets the filtered poses
// back.  One tick is ONE batched init + ONE batched step on the GPU instead of the reference's
// per-id loop.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "target_manager.hpp"

namespace te {

class MeasurementIngest {
 public:
  // type/Q/R/P0: what new targets are created with (RosTargetManager loads them from ROS
  // parameters, target_manager_ros.cpp:19-23); pass Q == nullptr to use the manager's defaults.
  MeasurementIngest(TargetManager* manager, int type, const double* Q, const double* R, const double* P0);

  void setTargetTokenName(const std::string& token) { token_name_ = token; }     // :94-97
  void setExpirationTime(double seconds) { expiration_time_ = seconds; }          // :99-103
  double time() const { return t_; }

  // Measurement::update (target_manager_ros.hpp:96-114): a stamp newer than the stored one marks
  // the mailbox "new" and records the time; an older or equal one clears the flag.  The pose is
  // stored either way.
  void push(unsigned id, double stamp, const double* pose7);
  // measurementCallBack for one transform (target_manager_ros.cpp:29-37): returns 1 if taken, 0 if
  // the frame name does not contain the token, -1 if it does but is not "<name>_<id>" (the
  // reference then stops processing the rest of the message).
  int push_named(const std::string& child_frame_id, double stamp, const double* pose7);

  // RosTargetManager::update(dt) (target_manager_ros.cpp:41-92).  `now` replaces ros::Time::now().
  // Fills ids (ascending) and their filtered poses [n][7]; returns n = live targets.
  long tick(double dt, double now, std::vector<unsigned>& ids_out, std::vector<double>& poses_out);

  size_t mailboxes() const { return measurements_.size(); }

 private:
  struct Mailbox {                 // class Measurement
    bool new_meas = true;          // ctor :78-82
    double last_meas_time = 0.0;
    double stamp = 0.0;            // tr_.header.stamp (default-constructed: 0)
    double pose[7] = {0, 0, 0, 0, 0, 0, 0};
  };
  TargetManager* manager_;
  bool use_defaults_;
  TargetManager::target_t type_;
  std::vector<double> Q_, R_, P_;
  std::string token_name_ = "target";   // target_manager_ros.cpp:9
  double t_ = 0.0;
  double expiration_time_ = 1000.0;     // "Dummy value", :11
  std::map<unsigned, Mailbox> measurements_;
};

// utils.hpp:273-313 splitString / getId: "xxx_id" -> id; exactly one '_' separated pair
bool parse_frame_id(const std::string& s, unsigned& id);

}  // namespace te
