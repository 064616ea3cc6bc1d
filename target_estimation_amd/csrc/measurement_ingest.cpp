// measurement_ingest.cpp -- see measurement_ingest.hpp.
#include "measurement_ingest.hpp"

#include <cstdlib>
#include <cstring>
#include <iostream>

namespace te {

bool parse_frame_id(const std::string& s, unsigned& id) {
  // splitString(s, "_") must give exactly two tokens (utils.hpp:302-313)
  const size_t first = s.find('_');
  if (first == std::string::npos) return false;
  if (s.find('_', first + 1) != std::string::npos) return false;
  const std::string tail = s.substr(first + 1);
  char* end = nullptr;
  const long v = std::strtol(tail.c_str(), &end, 10);   // std::stoi: leading digits, throws if none
  if (end == tail.c_str()) return false;
  id = (unsigned)v;
  return true;
}

MeasurementIngest::MeasurementIngest(TargetManager* manager, int type, const double* Q, const double* R, const double* P0)
    : manager_(manager), use_defaults_(Q == nullptr), type_((TargetManager::target_t)type) {
  if (!use_defaults_) {
    const int n = model_n(type), m = model_m(type);
    Q_.assign(Q, Q + n * n);
    R_.assign(R, R + m * m);
    P_.assign(P0, P0 + n * n);
  }
}

void MeasurementIngest::push(unsigned id, double stamp, const double* pose7) {
  Mailbox& mb = measurements_[id];          // operator[] default-constructs, as measurements_[id] does
  if (stamp > mb.stamp) {                   // :104-108
    mb.new_meas = true;
    mb.last_meas_time = stamp;
  } else {
    mb.new_meas = false;                    // :110-112
  }
  mb.stamp = stamp;                         // tr_ = tr, :113
  std::memcpy(mb.pose, pose7, sizeof(double) * 7);
}

int MeasurementIngest::push_named(const std::string& child_frame_id, double stamp, const double* pose7) {
  if (child_frame_id.find(token_name_) == std::string::npos) return 0;
  unsigned id = 0;
  if (!parse_frame_id(child_frame_id, id)) return -1;
  push(id, stamp, pose7);
  return 1;
}

long MeasurementIngest::tick(double dt, double now, std::vector<unsigned>& ids_out, std::vector<double>& poses_out) {
  std::vector<unsigned> ids, create_ids, expired;
  std::vector<double> meas, create_pose;
  std::vector<unsigned char> has;
  ids.reserve(measurements_.size());
  for (auto& kv : measurements_) {
    const unsigned id = kv.first;
    Mailbox& mb = kv.second;
    // Measurement::read succeeds whenever new_meas_ is set; it does NOT clear the flag
    // (target_manager_ros.hpp:84-94), so a mailbox keeps feeding its last pose until a message with
    // a non-newer stamp arrives.
    const bool got = mb.new_meas;
    if (got && !manager_->hasTarget(id)) {  // :54-58
      create_ids.push_back(id);
      create_pose.insert(create_pose.end(), mb.pose, mb.pose + 7);
    }
    ids.push_back(id);
    has.push_back(got ? 1 : 0);
    meas.insert(meas.end(), mb.pose, mb.pose + 7);
    if (mb.last_meas_time > 0.0 && (now - mb.last_meas_time) >= expiration_time_) expired.push_back(id);  // :67
  }
  if (!create_ids.empty()) {
    const long n = (long)create_ids.size();
    if (use_defaults_)
      manager_->initBatch(create_ids.data(), n, dt, t_, create_pose.data(), nullptr, nullptr);
    else
      manager_->initBatch(type_, create_ids.data(), n, dt, t_, Q_.data(), R_.data(), P_.data(), false, create_pose.data(),
                          nullptr, nullptr);
  }
  if (!ids.empty()) manager_->updateBatch(ids.data(), (long)ids.size(), dt, meas.data(), has.data());  // :60,:64
  std::vector<unsigned> to_erase;
  for (unsigned id : expired) {             // :67-72
    std::cerr << "Timeout for target " << id << std::endl;
    measurements_.erase(id);
    if (manager_->hasTarget(id)) to_erase.push_back(id);
  }
  if (!to_erase.empty()) manager_->eraseBatch(to_erase.data(), (long)to_erase.size());   // one compaction launch per batch
  ids_out = manager_->getAvailableTargets();  // :78
  poses_out.assign(ids_out.size() * 7, 0.0);
  if (!ids_out.empty())
    manager_->getPoseBatch(ids_out.data(), (long)ids_out.size(), poses_out.data(), nullptr, nullptr, nullptr);
  t_ = t_ + dt;                             // :89
  manager_->log();                          // :91
  return (long)ids_out.size();
}

}  // namespace te
