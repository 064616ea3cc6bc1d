/*
 * ref_stream.cpp -- the measurement stream of the reference's integration test, regenerated
 * with the same libstdc++ generators.  TEST INFRASTRUCTURE ONLY (see te_oracle.h).
 *
 * Restates test/target_manager_test.cpp:11-20 (noise parameters, static default-seeded
 * std::default_random_engine shared by the four TESTs, _n_points, goal, omega) and
 * generateLinearMeasurements :82-115.  gtest runs the TESTs in definition order
 * (UniformVelocity, UniformAcceleration, AngularRates, AngularVelocities: :148,:192,:236,:289),
 * so model k consumes draws [3*n*k, 3*n*(k+1)) of the one generator.
 */
#include <random>
#include "te_oracle.h"

extern "C" void orc_ref_test_stream(double* meas /*[n_models][n_points][7]*/, int n_models,
                                    int n_points, double dt, const double* goal3,
                                    const double* omega3) {
  std::default_random_engine generator;                       // :14, default seed
  std::normal_distribution<double> normal_dist(0.0, 0.01);    // :11-12,:15
  for (int k = 0; k < n_models; ++k) {
    double* mp = meas + (long)k * n_points * 7;
    double q[4] = {0.0, 0.0, 0.0, 1.0};                       // Quaterniond::Identity(), [x y z w]
    double M[16];
    orc_qtran_f64(dt, omega3, M);
    for (int i = 0; i < n_points; ++i) {
      for (int c = 0; c < 3; ++c) {
        // Eigen::VectorXd::LinSpaced(n, 0, goal): low + i*step, last element == high (:92-94)
        const double step = (goal3[c] - 0.0) / (double)(n_points - 1);
        const double real = (i == n_points - 1) ? goal3[c] : 0.0 + (double)i * step;
        mp[i * 7 + c] = real + normal_dist(generator);        // :102-104, draw order x,y,z
      }
      for (int c = 0; c < 4; ++c) mp[i * 7 + 3 + c] = q[c];    // :106-109
      double qn[4];
      for (int r = 0; r < 4; ++r) {                           // :112
        double acc = M[r * 4] * q[0];
        for (int c = 1; c < 4; ++c) acc += M[r * 4 + c] * q[c];
        qn[r] = acc;
      }
      orc_quat_normalize_f64(qn);                             // :113
      for (int c = 0; c < 4; ++c) q[c] = qn[c];
    }
  }
}
