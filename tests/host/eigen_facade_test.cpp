// eigen_facade_test.cpp -- include/target_estimation_amd/target_manager_eigen.hpp through a compiler and against the C ABI.
//
// Built with g++ against tests/host/eigen_standin (a stand-in for the Eigen members the facade touches: Eigen3 is absent
// from the image) and linked with libtarget_estimation_amd.so.  Mirrors how the reference's integration test drives the
// plugin (test/target_manager_test.cpp:139-158: init -> update(id, dt, meas) -> getTargetPose / getTarget(id)) and checks
// every facade call against the same call made directly on the C symbols with row-major arrays -- in particular that a
// NON-symmetric column-major MatrixXd reaches the library transposed correctly and that getP() comes back in (row, col)
// order.  Usage: eigen_facade_test <model yaml>   (needs a GPU; `--syntax-only` builds are the CPU check)
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "target_manager_eigen.hpp"

#ifndef TARGET_ESTIMATION_AMD_HAS_EIGEN
#error "the facade is inert: <Eigen/Dense> was not found on the include path"
#endif

using target_estimation_amd::TargetManager;


static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); ++fails; } } while (0)

int main(int argc, char** argv) {
  if (argc < 2) { std::printf("usage: %s model.yaml\n", argv[0]); return 2; }
  const int n = 6, m = 3;   // uniform_velocity
  // a coupled, NON-symmetric Q / P0 (legal input for the reference: it never checks), so that a transposition shows
  Eigen::MatrixXd Q(n, n), P0(n, n), R(m, m);
  double q_rm[36], p_rm[36], r_rm[9];
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      Q(i, j) = (i == j ? 1e-3 : 0.0) + 1e-5 * (i + 1) - 2e-6 * j;
      P0(i, j) = (i == j ? 1.0 : 0.0) + 1e-2 * i - 3e-3 * j;
      q_rm[i * n + j] = Q(i, j);
      p_rm[i * n + j] = P0(i, j);
    }
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < m; ++j) { R(i, j) = (i == j ? 1e-2 : 0.0) + 1e-4 * i - 5e-5 * j; r_rm[i * m + j] = R(i, j); }
  CHECK(Q.data()[1] == Q(1, 0));   // the stand-in is column-major, as Eigen's default

  Eigen::Vector7d p0 = Eigen::Vector7d::Zero();
  p0(0) = 0.3; p0(1) = -0.2; p0(2) = 1.1; p0(6) = 1.0;
  Eigen::Vector6d v0 = Eigen::Vector6d::Zero();
  v0(0) = 0.5; v0(1) = 0.1;
  const double dt = 0.01;

  // (1) the facade
  TargetManager mgr(argv[1]);
  TargetManager::target_t type;
  CHECK(mgr.selectTargetType("uniform_velocity", type) && type == TargetManager::UNIFORM_VELOCITY);
  CHECK(!mgr.selectTargetType("nonsense", type));
  mgr.init(type, 11, dt, 0.0, Q, R, P0, p0, v0);
  mgr.init(12, dt, 0.0, p0);                       // default model of the YAML
  // (2) the same through the C symbols, row-major arrays
  target_manager_c* c = target_manager_new(argv[1]);
  CHECK(c != nullptr);
  CHECK(target_manager_init_typed(c, (int)type, 11, dt, 0.0, q_rm, r_rm, p_rm, p0.data(), v0.data(), nullptr) >= 0);

  for (int s = 0; s < 40; ++s) {
    Eigen::Vector7d meas = p0;
    meas(0) += 0.5 * dt * (s + 1) + 1e-3 * std::sin(0.7 * s);
    meas(1) += 0.1 * dt * (s + 1);
    if (s % 5 == 4) {
      CHECK(mgr.update(11, dt));                                     // predict only
      unsigned id = 11;
      CHECK(target_manager_update_meas_batch(c, &id, 1, dt, nullptr, nullptr) == 1);
    } else {
      CHECK(mgr.update(11, dt, meas));
      target_manager_update_meas(c, 11, dt, meas.data());   // one of the reference's ten symbols
    }
    CHECK(mgr.update(12, dt, meas));
  }
  CHECK(!mgr.update(99, dt));   // unknown id: false, as the reference

  // state and covariance: facade (Eigen types) vs C ABI (row-major)
  auto h = mgr.getTarget(11);
  CHECK((bool)h);
  CHECK(!mgr.getTarget(99));
  if (h) {
    const Eigen::VectorXd x = h->getEstimator()->getState();
    const Eigen::MatrixXd P = h->getEstimator()->getP();
    double xb[18], Pb[18 * 18];
    unsigned id = 11;
    CHECK(target_manager_get_state_batch(c, &id, 1, xb, Pb) == n);
    CHECK(x.size() == n && P.rows() == n && P.cols() == n);
    double asym = 0;
    for (int i = 0; i < n; ++i) {
      CHECK(x(i) == xb[i]);
      for (int j = 0; j < n; ++j) {
        CHECK(P(i, j) == Pb[i * n + j]);       // bitwise: same kernels, same inputs -- unless Q / P0 went in transposed
        asym = std::fmax(asym, std::fabs(Pb[i * n + j] - Pb[j * n + i]));
      }
    }
    CHECK(asym > 0);   // the input really was non-symmetric, so the comparison above can see a transposition
    Eigen::Vector7d pose, pose_c;
    CHECK(mgr.getTargetPose(11, pose));
    CHECK(target_manager_get_est_pose(c, 11, pose_c.data()));
    for (int i = 0; i < 7; ++i) CHECK(pose(i) == pose_c(i) && h->getEstimatedPose()(i) == pose_c(i));
    Eigen::Vector6d tw, tw_c;
    CHECK(mgr.getTargetTwist(11, tw));
    CHECK(target_manager_get_est_twist(c, 11, tw_c.data()));
    for (int i = 0; i < 6; ++i) CHECK(tw(i) == tw_c(i) && h->getEstimatedTwist()(i) == tw_c(i));
    CHECK(std::fabs(tw(0) - 0.5) < 0.05);     // the reference test's own kind of bound (target_manager_test.cpp:158)
    // extrapolation getters
    const Eigen::Vector7d ahead = h->getEstimatedPose(h->getTime() + 0.1);
    CHECK(std::fabs(ahead(0) - (pose(0) + 0.1 * tw(0))) < 1e-12);
    CHECK(h->getNumberMeasurements() == 32 && mgr.getNumberMeasurements(11) == 32);
    CHECK(std::fabs(h->getTime() - 40 * dt) < 1e-12);
    const Eigen::Vector6d acc = h->getEstimatedAcceleration();
    for (int i = 0; i < 6; ++i) CHECK(acc(i) == 0.0);   // uniform_velocity has no acceleration state
  }
  const std::vector<unsigned int> ids = mgr.getAvailableTargets();
  CHECK(ids.size() == 2 && ids[0] == 11 && ids[1] == 12);
  Eigen::Vector3d origin;
  origin(0) = 0; origin(1) = 0; origin(2) = 0;
  CHECK(mgr.getIntersectionTimeWithSphere(11, 0.5, origin, 10.0) == -1.0);   // uniform_velocity never intersects (intersection_solver.cpp:61-63)
  // the round-3 getters of TargetInterface / the estimator (target_interface.hpp:94-148, kalman.hpp:74-89)
  if (h) {
    double per = 0, T16[16], q_back[36], r_back[9], p_back[36];
    CHECK(target_manager_get_period_estimate(c, 11, &per) && h->getPeriodEstimate() == per && per == -1.0);   // no angular rate in this model
    CHECK(target_manager_get_estimated_transform(c, 11, T16));
    const Eigen::Isometry3d T = h->getEstimatedTransform();
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) CHECK(T.matrix()(i, j) == T16[i * 4 + j]);
    CHECK(T.matrix()(0, 3) == h->getEstimatedPose()(0) && T.matrix()(0, 0) == 1.0 && T.matrix()(3, 3) == 1.0);
    CHECK(h->getN() == 6 && h->getM() == 3);
    CHECK(target_manager_get_model_matrices(c, 11, q_back, r_back, p_back));
    const Eigen::MatrixXd Qb = h->getEstimator()->getQ(), Rb = h->getEstimator()->getR(), Pb0 = h->getEstimator()->getP0();
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) CHECK(Qb(i, j) == Q(i, j) && Pb0(i, j) == P0(i, j) && q_back[i * n + j] == Q(i, j));
    for (int i = 0; i < m; ++i)
      for (int j = 0; j < m; ++j) CHECK(Rb(i, j) == R(i, j));
    const Eigen::Vector7d mp = h->getMeasuredPose();     // not kept by default: the initial pose
    CHECK(mp(6) == 1.0 && mp(0) == 0.0);
  }
  // IntersectionSolver with the reference's constructor and method signatures (intersection_solver.hpp:63,73,86)
  {
    TargetManager::Ptr shared(new TargetManager(argv[1]));
    shared->setKeepMeasurement(true);
    Eigen::MatrixXd Qa = Eigen::MatrixXd::Zero(9, 9), Pa = Eigen::MatrixXd::Zero(9, 9), Ra = Eigen::MatrixXd::Zero(3, 3);
    for (int i = 0; i < 9; ++i) { Qa(i, i) = 1e-6; Pa(i, i) = 1e-2; }
    for (int i = 0; i < 3; ++i) Ra(i, i) = 1e-4;
    Eigen::Vector7d start = Eigen::Vector7d::Zero();
    start(0) = 10.0; start(6) = 1.0;
    Eigen::Vector6d vin = Eigen::Vector6d::Zero(), ain = Eigen::Vector6d::Zero();
    // inbound and decelerating: x(d) = 10 - 5 d + d^2 / 2 crosses +-2 at d = 2, 4, 6, 8 -- every real root of the quartic is
    // positive (the reference takes the smallest real root BY VALUE and answers -1 if that is negative, intersection_solver.cpp:83)
    vin(0) = -5.0; ain(0) = 1.0;
    shared->init(TargetManager::UNIFORM_ACCELERATION, 5, dt, 0.0, Qa, Ra, Pa, start, vin, ain);
    target_estimation_amd::IntersectionSolver solver(shared, 3), solver_default(shared);
    Eigen::Vector7d ip;
    bool conv = false, conv_any = false;
    for (int s = 0; s < 12; ++s) {
      const double t = dt * (s + 1);
      Eigen::Vector7d z = start;
      z(0) = 10.0 - 5.0 * t + 0.5 * t * t;
      CHECK(shared->update(5, dt, z));
      const double d = solver.getIntersectionTimeWithSphere(5, t, origin, 2.0);
      CHECK(std::fabs(d - (2.0 - t)) < 0.05 && d == target_manager_get_intersection_time_with_sphere(shared->handle(), 5, t, origin.data(), 2.0));
      conv = solver.getIntersectionPoseWithSphere(5, t, 0.05, 0.05, origin, 2.0, ip);
      conv_any = conv_any || conv;
      CHECK(std::fabs(std::sqrt(ip(0) * ip(0) + ip(1) * ip(1) + ip(2) * ip(2)) - 2.0) < 1e-6);   // the pose is ON the sphere
      if (s == 0) CHECK(!conv);                          // first pose vs initPose: 2 m apart
    }
    CHECK(conv_any && conv);                             // the intersection point settles: the filtered error falls below 5 cm
    CHECK(!solver_default.getIntersectionPoseWithSphere(99, 0.1, 0.05, 0.05, origin, 2.0, ip) && ip(6) == 1.0 && ip(0) == 0.0);
    const Eigen::Vector7d last = shared->getTarget(5)->getMeasuredPose();
    CHECK(last(0) == 10.0 - 5.0 * (dt * 12) + 0.5 * (dt * 12) * (dt * 12));
  }
  mgr.update(dt);   // all targets, predict only
  CHECK(std::fabs(mgr.getTarget(12)->getTime() - 41 * dt) < 1e-12);
  CHECK(mgr.erase(12) && !mgr.erase(12));
  target_manager_delete(c);
  if (fails) { std::printf("%d checks failed\n", fails); return 1; }
  std::printf("eigen facade test ok\n");
  return 0;
}
