// target_manager_eigen.hpp -- Eigen-typed, source-compatible facade of the reference's C++ plugin
// surface over the C ABI of libtarget_estimation_amd.so.
//
// The reference's in-process callers (RosTargetManager : public TargetManager,
// target_manager_ros.hpp:136; IntersectionSolver, intersection_solver.cpp:45-49,104; the
// integration test, target_manager_test.cpp:139-144,158) use `TargetManager` and
// `TargetInterface` with Eigen arguments (include/target_estimation/target_manager.hpp:43-203,
// target_interface.hpp:58-160).  This header gives them the same class and method names with the
// same argument types; every call forwards to the C symbols of target_manager_c.h /
// target_batch_c.h, so the filters run on the GPU.  Header-only, needs only <Eigen/Dense> and the
// two C headers (no HIP headers).
//
// Eigen3 is absent from the authoring environment (SURVEY.md headline 3), so this header has never met the real
// library.  What it has met: a test-only stand-in for the Eigen members it uses (tests/host/eigen_standin, Eigen's
// storage-order semantics) under g++ -Wall -Wextra -Werror, and tests/host/eigen_facade_test.cpp on the GPU -- every
// call below against the same call on the C symbols, bitwise, with non-symmetric column-major Q / P0
// (tests/test_eigen_facade.py).  It is guarded so that it is inert where Eigen is missing.  Differences from the reference types:
// getTarget(id) returns a small value handle (the filter state lives in HBM, not in a host object);
// MatrixXd arguments are converted to the row-major arrays the C ABI takes.
#pragma once
#if defined(__has_include)
#if __has_include(<Eigen/Dense>)
#define TARGET_ESTIMATION_AMD_HAS_EIGEN 1
#endif
#endif

#ifdef TARGET_ESTIMATION_AMD_HAS_EIGEN
#include <Eigen/Dense>
#include <memory>
#include <string>
#include <vector>

#include "target_batch_c.h"

namespace Eigen {  // include/target_estimation/utils.hpp:50-57
typedef Matrix<double, 6, 1> Vector6d;
typedef Matrix<double, 7, 1> Vector7d;
}  // namespace Eigen

namespace target_estimation_amd {

typedef Eigen::Matrix<double, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor> RowMatrixXd;

// What getTarget(id)->getEstimator() exposes in the reference (kalman.hpp:69-89)
class EstimatorView {
 public:
  EstimatorView(target_manager_c* m, unsigned id) : m_(m), id_(id) {}
  Eigen::VectorXd getState() const { Eigen::VectorXd x; Eigen::MatrixXd P; fetch(x, P); return x; }
  Eigen::MatrixXd getP() const { Eigen::VectorXd x; Eigen::MatrixXd P; fetch(x, P); return P; }
  Eigen::MatrixXd getQ() const { return model(0); }    // kalman.hpp:74
  Eigen::MatrixXd getR() const { return model(1); }    // kalman.hpp:79
  Eigen::MatrixXd getP0() const { return model(2); }   // kalman.hpp:89

 private:
  Eigen::MatrixXd model(int which) const {   // 0 Q, 1 R, 2 P0: the matrices the target was created with
    const int n = target_manager_get_n(m_, id_), m = target_manager_get_m(m_, id_);
    Eigen::MatrixXd out;
    if (n <= 0) return out;
    double Q[18 * 18], R[6 * 6], P0[18 * 18];
    if (!target_manager_get_model_matrices(m_, id_, which == 0 ? Q : nullptr, which == 1 ? R : nullptr, which == 2 ? P0 : nullptr)) return out;
    if (which == 1) out = Eigen::Map<RowMatrixXd>(R, m, m);
    else out = Eigen::Map<RowMatrixXd>(which == 0 ? Q : P0, n, n);
    return out;
  }
  void fetch(Eigen::VectorXd& x, Eigen::MatrixXd& P) const {
    double xb[18], Pb[18 * 18];
    const long n = target_manager_get_state_batch(m_, &id_, 1, xb, Pb);
    if (n <= 0) { x.resize(0); P.resize(0, 0); return; }
    x = Eigen::Map<Eigen::VectorXd>(xb, n);
    P = Eigen::Map<RowMatrixXd>(Pb, n, n);
  }
  target_manager_c* m_;
  unsigned id_;
};

// The subset of TargetInterface (target_interface.hpp:58-160) reachable through getTarget(id)
class TargetHandle {
 public:
  typedef std::shared_ptr<TargetHandle> Ptr;
  TargetHandle(target_manager_c* m, unsigned id) : m_(m), id_(id), est_(m, id) {}
  unsigned int getID() const { return id_; }
  Eigen::Vector7d getEstimatedPose() const { Eigen::Vector7d p; target_manager_get_est_pose(m_, id_, p.data()); return p; }
  Eigen::Vector6d getEstimatedTwist() const { Eigen::Vector6d v; target_manager_get_est_twist(m_, id_, v.data()); return v; }
  Eigen::Vector6d getEstimatedAcceleration() const { Eigen::Vector6d a; target_manager_get_est_acceleration(m_, id_, a.data()); return a; }
  Eigen::Vector7d getEstimatedPose(const double& t) const { Eigen::Vector7d p; at(t, p.data(), nullptr, nullptr); return p; }
  Eigen::Vector6d getEstimatedTwist(const double& t) const { Eigen::Vector6d v; at(t, nullptr, v.data(), nullptr); return v; }
  Eigen::Vector6d getEstimatedAcceleration(const double& t) const { Eigen::Vector6d a; at(t, nullptr, nullptr, a.data()); return a; }
  double getTime() const { double t = 0; target_manager_get_time(m_, id_, &t); return t; }
  // target_interface.hpp:94: 2 pi / |omega|, -1 if the target is not rotating
  double getPeriodEstimate() const { double p = -1.0; target_manager_get_period_estimate(m_, id_, &p); return p; }
  // target_interface.hpp:106
  Eigen::Isometry3d getEstimatedTransform() const {
    double T[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    target_manager_get_estimated_transform(m_, id_, T);
    Eigen::Isometry3d out;
    out.matrix() = Eigen::Map<Eigen::Matrix<double, 4, 4, Eigen::RowMajor> >(T, 4, 4);
    return out;
  }
  // target_interface.hpp:130: the last measurement ([0 0 0 0 0 0 1] before the first).  The manager keeps it only on
  // request (target_manager_set_keep_measurement / TargetManager::setKeepMeasurement below).
  Eigen::Vector7d getMeasuredPose() const {
    Eigen::Vector7d p = Eigen::Vector7d::Zero();
    p(6) = 1.0;
    target_manager_get_measured_pose(m_, id_, p.data());
    return p;
  }
  unsigned int getN() const { return (unsigned int)target_manager_get_n(m_, id_); }   // target_interface.hpp:142
  unsigned int getM() const { return (unsigned int)target_manager_get_m(m_, id_); }   // :148
  long long getNumberMeasurements() const { return target_manager_get_n_measurements(m_, id_); }
  const EstimatorView* getEstimator() const { return &est_; }

 private:
  void at(double t, double* p, double* v, double* a) const { target_manager_get_est_at_batch(m_, &id_, 1, t, p, v, a, nullptr); }
  target_manager_c* m_;
  unsigned id_;
  EstimatorView est_;
};

class TargetManager {
 public:
  typedef std::shared_ptr<TargetManager> Ptr;
  enum target_t { ANGULAR_RATES = 0, ANGULAR_VELOCITIES, UNIFORM_ACCELERATION, UNIFORM_VELOCITY };  // target_manager.hpp:38

  TargetManager() : m_(target_manager_new_ex(nullptr, TARGET_DTYPE_F64, 0)) { if (!m_) throw "TargetManager constructor failed!"; }
  explicit TargetManager(const std::string& file) : m_(target_manager_new(file.c_str())) {
    if (!m_) throw "TargetManager default constructor failed!";  // target_manager.cpp:114-115
  }
  virtual ~TargetManager() { target_manager_delete(m_); }
  TargetManager(const TargetManager&) = delete;
  TargetManager& operator=(const TargetManager&) = delete;

  // target_manager.hpp:75-76
  void init(const unsigned int& id, const double& dt0, const double& t0, const Eigen::Vector7d& p0,
            const Eigen::Vector6d& v0 = Eigen::Vector6d::Zero(), const Eigen::Vector6d& a0 = Eigen::Vector6d::Zero()) {
    if (target_manager_init_batch(m_, &id, 1, dt0, t0, p0.data(), v0.data(), a0.data()) < 0)
      throw "TargetManager::init failed, can not find default values to load!";  // target_manager.cpp:141
  }
  // target_manager.hpp:85-87
  void init(const target_t& type, const unsigned int& id, const double& dt0, const double& t0, const Eigen::MatrixXd& Q,
            const Eigen::MatrixXd& R, const Eigen::MatrixXd& P0, const Eigen::Vector7d& p0,
            const Eigen::Vector6d& v0 = Eigen::Vector6d::Zero(), const Eigen::Vector6d& a0 = Eigen::Vector6d::Zero()) {
    const RowMatrixXd q = Q, r = R, p = P0;
    target_manager_init_typed(m_, (int)type, id, dt0, t0, q.data(), r.data(), p.data(), p0.data(), v0.data(), a0.data());
  }
  bool update(const unsigned int& id, const double& dt, const Eigen::Vector7d& meas) {  // target_manager.hpp:106
    return target_manager_update_meas_batch(m_, &id, 1, dt, meas.data(), nullptr) == 1;
  }
  bool update(const unsigned int& id, const double& dt) {                               // :114
    return target_manager_update_meas_batch(m_, &id, 1, dt, nullptr, nullptr) == 1;
  }
  virtual void update(const double& dt) { target_manager_update_all(m_, dt); }          // :120
  bool erase(const unsigned int& id) { return target_manager_erase(m_, id) == 1; }     // :127
  TargetHandle::Ptr getTarget(const unsigned int& id) {                                 // :134
    unsigned char found = 0;
    target_manager_get_est_batch(m_, &id, 1, nullptr, nullptr, nullptr, &found);
    return found ? std::make_shared<TargetHandle>(m_, id) : nullptr;
  }
  bool getTargetPose(const unsigned int& id, Eigen::Vector7d& pose) { return target_manager_get_est_pose(m_, id, pose.data()); }
  bool getTargetTwist(const unsigned int& id, Eigen::Vector6d& twist) { return target_manager_get_est_twist(m_, id, twist.data()); }
  bool getTargetAcceleration(const unsigned int& id, Eigen::Vector6d& acc) { return target_manager_get_est_acceleration(m_, id, acc.data()); }
  long long getNumberMeasurements(const unsigned int& id) { return target_manager_get_n_measurements(m_, id); }
  void log() { target_manager_log(m_); }
  std::vector<unsigned int> getAvailableTargets() {
    std::vector<unsigned int> ids((size_t)target_manager_size(m_));
    if (!ids.empty()) target_manager_get_available_targets(m_, ids.data(), (long)ids.size());
    return ids;
  }
  bool selectTargetType(const std::string& s, target_t& type) {  // target_manager.cpp:52-65
    if (s == "angular_rates") type = ANGULAR_RATES;
    else if (s == "angular_velocities") type = ANGULAR_VELOCITIES;
    else if (s == "uniform_acceleration") type = UNIFORM_ACCELERATION;
    else if (s == "uniform_velocity") type = UNIFORM_VELOCITY;
    else return false;
    return true;
  }
  // IntersectionSolver::getIntersectionTimeWithSphere / PoseWithSphere (intersection_solver.hpp:84-101)
  double getIntersectionTimeWithSphere(const unsigned int& id, const double& t1, const Eigen::Vector3d& origin, const double& radius) {
    return target_manager_get_intersection_time_with_sphere(m_, id, t1, origin.data(), radius);
  }
  void setKeepMeasurement(bool on) { target_manager_set_keep_measurement(m_, on ? 1 : 0); }   // see TargetHandle::getMeasuredPose
  target_manager_c* handle() { return m_; }

 protected:
  target_manager_c* m_;
};

// The reference's IntersectionSolver (include/target_estimation/intersection_solver.hpp:56-126): same constructor, same two
// methods with the same Eigen argument types, ONE convergence gate per solver object (src/intersection_solver.cpp:19-40,
// 91-124) -- over the C entry points target_intersection_solver_*.
class IntersectionSolver {
 public:
  typedef std::shared_ptr<IntersectionSolver> Ptr;
  IntersectionSolver(TargetManager::Ptr target_manager, const unsigned int filters_length = 250)   // :63
      : target_manager_(target_manager), s_(target_intersection_solver_new(target_manager ? target_manager->handle() : nullptr, filters_length)) {
    if (!s_) throw "IntersectionSolver constructor failed!";
  }
  ~IntersectionSolver() { target_intersection_solver_delete(s_); }
  IntersectionSolver(const IntersectionSolver&) = delete;
  IntersectionSolver& operator=(const IntersectionSolver&) = delete;
  // :73
  double getIntersectionTimeWithSphere(const unsigned int& id, const double& t1, const Eigen::Vector3d& origin, const double& radius) {
    return target_intersection_solver_get_time_with_sphere(s_, id, t1, origin.data(), radius);
  }
  // :86
  bool getIntersectionPoseWithSphere(const unsigned int& id, const double& t1, const double& pos_th, const double& ang_th,
                                     const Eigen::Vector3d& origin, const double& radius, Eigen::Vector7d& intersection_pose) {
    return target_intersection_solver_get_pose_with_sphere(s_, id, t1, pos_th, ang_th, origin.data(), radius, intersection_pose.data());
  }

 private:
  TargetManager::Ptr target_manager_;   // keeps the manager alive, as the reference's member does
  target_intersection_solver_c* s_;
};

}  // namespace target_estimation_amd
#endif  // TARGET_ESTIMATION_AMD_HAS_EIGEN
