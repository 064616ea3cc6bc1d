/* node_tick.c -- the reference node's tick in two calls, from plain C (gcc, no HIP headers, no Python).
 *
 * The reference's node steps its targets one by one every tick -- update(id, dt, measurement) for a target whose transform
 * arrived, update(id, dt) for the others (src/target_manager_ros.cpp:41-64) -- and then reads every target's pose to publish
 * it (:78-87).  With this library the same tick is ONE target_manager_update_meas_batch (ids in any order, a mask for "no
 * measurement this tick") and ONE target_manager_get_est_batch.  At node sizes (up to a thousand ids per call) both go through
 * the one-target queue and the host-resident getter table: one launch per tick -- from C 11 - 14 us for 40 targets, 19 us for
 * 300, 33 us for 1000 (this program on one MI355X, profiles/r04_fused_flush.txt).  The program checks the two-call tick against the reference's own call pattern on a
 * second manager (the ten symbols, target by target): same poses to the last bit.
 *
 *   gcc -std=c99 -O2 -I include/target_estimation_amd examples/node_tick.c -o node_tick \
 *       -L target_estimation_amd/lib -ltarget_estimation_amd -lm -Wl,-rpath,$PWD/target_estimation_amd/lib
 *   ./node_tick models/model_angular_velocities_params.yaml 40 500 */
#define _POSIX_C_SOURCE 199309L
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "target_batch_c.h"
#include "target_manager_c.h"

static double now_us(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return 1e6 * (double)t.tv_sec + 1e-3 * (double)t.tv_nsec;
}

/* a target on a circle of its own, seen through a little noise */
static void measurement(unsigned id, int tick, double* m) {
  const double w = 0.3 + 0.01 * (double)(id % 17), r = 1.0 + 0.05 * (double)(id % 5), t = 0.004 * (double)tick;
  unsigned s = id * 2654435761u + (unsigned)tick * 40503u;
  double noise[3];
  int k;
  for (k = 0; k < 3; ++k) { s = s * 1664525u + 1013904223u; noise[k] = 1e-3 * ((double)(s >> 8) / 16777216.0 - 0.5); }
  m[0] = r * cos(w * t) + noise[0]; m[1] = r * sin(w * t) + noise[1]; m[2] = 0.1 * (double)(id % 3) + noise[2];
  m[3] = 0.0; m[4] = 0.0; m[5] = sin(0.5 * w * t); m[6] = cos(0.5 * w * t);   /* yaw = w t */
}

int main(int argc, char** argv) {
  const char* model = argc > 1 ? argv[1] : "models/model_angular_velocities_params.yaml";
  const long n = argc > 2 ? atol(argv[2]) : 40;
  const int ticks = argc > 3 ? atoi(argv[3]) : 500;
  const double dt = 0.004;
  long i;
  int s, k;
  if (n < 1 || n > 1024 || ticks < 1) { fprintf(stderr, "1 .. 1024 targets\n"); return 2; }
  target_manager_c* batch = target_manager_new(model);   /* the two-call tick */
  target_manager_c* loop = target_manager_new(model);    /* the reference's call pattern */
  if (!batch || !loop) { fprintf(stderr, "no manager (model file? GPU?)\n"); return 3; }
  unsigned* ids = (unsigned*)malloc(sizeof(unsigned) * (size_t)n);
  double* meas = (double*)malloc(sizeof(double) * 7 * (size_t)n);
  unsigned char* has = (unsigned char*)malloc((size_t)n);
  double* pose = (double*)malloc(sizeof(double) * 7 * (size_t)n);
  unsigned char* found = (unsigned char*)malloc((size_t)n);
  for (i = 0; i < n; ++i) {
    ids[i] = 100u + 7u * (unsigned)i;                     /* frame ids as the node parses them: not dense, not sorted by arrival */
    measurement(ids[i], 0, meas + 7 * i);
    target_manager_init(loop, ids[i], dt, meas + 7 * i, 0.0);
  }
  if (target_manager_init_batch(batch, ids, n, dt, 0.0, meas, NULL, NULL) != n) return 4;

  double t_batch = 0.0, t_loop = 0.0, worst = 0.0;
  long missing = 0;
  for (s = 1; s <= ticks; ++s) {
    for (i = 0; i < n; ++i) {
      measurement(ids[i], s, meas + 7 * i);
      has[i] = (unsigned char)(((ids[i] + 3u * (unsigned)s) % 10u) != 0u);   /* one tick in ten a target's transform does not arrive */
    }
    double t0 = now_us();
    const long stepped = target_manager_update_meas_batch(batch, ids, n, dt, meas, has);
    const long read = target_manager_get_est_batch(batch, ids, n, pose, NULL, NULL, found);
    double t1 = now_us();
    if (stepped != n || read != n) return 5;
    if (s > 20) t_batch += t1 - t0;
    /* the reference's loop on the second manager */
    t0 = now_us();
    for (i = 0; i < n; ++i) {
      if (has[i]) target_manager_update_meas(loop, ids[i], dt, meas + 7 * i);
      else target_manager_update(loop, ids[i], dt);
    }
    for (i = 0; i < n; ++i) {
      double p[7];
      if (!target_manager_get_est_pose(loop, ids[i], p)) { ++missing; continue; }
      for (k = 0; k < 7; ++k) {
        const double d = fabs(p[k] - pose[7 * i + k]);
        if (d > worst) worst = d;
      }
      if (!found[i]) ++missing;
    }
    t1 = now_us();
    if (s > 20) t_loop += t1 - t0;
  }
  const int timed = ticks > 20 ? ticks - 20 : 1;
  printf("%ld targets, %d ticks: two calls per tick %.1f us, the ten symbols target by target %.1f us per tick\n", n, ticks, t_batch / timed, t_loop / timed);
  printf("largest difference between the two managers' poses: %.3g; missing: %ld\n", worst, missing);
  const int ok = worst == 0.0 && missing == 0;
  printf(ok ? "node tick ok\n" : "NODE TICK FAILED\n");
  target_manager_delete(batch);
  target_manager_delete(loop);
  free(ids); free(meas); free(has); free(pose); free(found);
  return ok ? 0 : 1;
}
