/* drop_in_test.c -- a plain C caller of the reference's C wrapper, compiled with gcc against
 * include/target_estimation_amd/target_manager_c.h and linked with libtarget_estimation_amd.so.
 * It uses ONLY the ten symbols of the reference's target_manager_c.h (plus nothing else), i.e. it
 * would compile unchanged against the reference's libtarget_c.  Straight-line target, noisy
 * measurements, uniform-velocity model: the filter must converge like the reference's test
 * (test/target_manager_test.cpp:179-189: end position within 1 cm, mean velocity within 0.01).
 * usage: drop_in_test <model.yaml>   (exit code 0 = pass) */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "target_manager_c.h"

static double noise(unsigned* s) {   /* small deterministic zero-mean noise, sigma ~ 0.01 */
  double acc = 0.0;
  for (int k = 0; k < 12; ++k) {
    *s = *s * 1664525u + 1013904223u;
    acc += (double)(*s >> 8) / 16777216.0;
  }
  return (acc - 6.0) * 0.01;
}

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s model.yaml\n", argv[0]); return 2; }
  target_manager_c* m = target_manager_new(argv[1]);
  if (!m) { fprintf(stderr, "target_manager_new failed\n"); return 3; }
  const int n_points = 4000;
  const double dt = 1.0 / 250.0, goal[3] = {0.2, 0.3, 0.4};
  double p0[7] = {0, 0, 0, 0, 0, 0, 1}, meas[7] = {0, 0, 0, 0, 0, 0, 1}, pose[7], twist[6], acc[6];
  unsigned seed = 12345u;
  target_manager_init(m, 7u, dt, p0, 0.0);
  target_manager_init(m, 7u, dt, p0, 0.0);            /* duplicate: message, no change */
  double vsum[3] = {0, 0, 0};
  for (int i = 0; i < n_points; ++i) {
    for (int c = 0; c < 3; ++c) meas[c] = goal[c] * (double)i / (double)(n_points - 1) + noise(&seed);
    target_manager_update_meas(m, 7u, dt, meas);
    if (!target_manager_get_est_pose(m, 7u, pose) || !target_manager_get_est_twist(m, 7u, twist)) return 4;
    for (int c = 0; c < 3; ++c) vsum[c] += twist[c];
  }
  target_manager_update(m, 7u, dt);                   /* one predict-only step */
  if (!target_manager_get_est_acceleration(m, 7u, acc)) return 5;
  int fail = 0;
  for (int c = 0; c < 3; ++c) {
    if (fabs(pose[c] - goal[c]) > 0.01) { fprintf(stderr, "end position %d off: %g\n", c, pose[c]); fail = 1; }
    if (fabs(vsum[c] / n_points - goal[c] / (n_points * dt)) > 0.01) { fprintf(stderr, "mean velocity %d off\n", c); fail = 1; }
  }
  if (pose[3] != 0.0 || pose[4] != 0.0 || pose[5] != 0.0 || pose[6] != 1.0) { fprintf(stderr, "quaternion\n"); fail = 1; }
  if (target_manager_get_n_measurements(m, 7u) != n_points) { fprintf(stderr, "n_measurements\n"); fail = 1; }
  if (target_manager_get_est_pose(m, 8u, pose)) { fprintf(stderr, "unknown id returned true\n"); fail = 1; }
  if (target_manager_get_n_measurements(m, 8u) != 0) fail = 1;
  target_manager_update_meas(m, 8u, dt, meas);        /* unknown id: message only */
  target_manager_log(m);
  target_manager_delete(m);
  printf(fail ? "DROP-IN TEST FAILED\n" : "drop-in test ok\n");
  return fail;
}
