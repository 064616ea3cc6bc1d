/* scalar_abi_rate.c -- cost of the reference's per-id C ABI for a C caller at the reference's scale:
 * T targets, each tick = T x target_manager_update_meas, then T x (get_est_pose + get_est_twist).
 * Also the interleaved pattern of the reference's own test (update one target, read it back at once).
 * build: gcc -O2 -I include/target_estimation_amd tools/scalar_abi_rate.c -o tools/_build/scalar_abi_rate
 *        -L target_estimation_amd/lib -ltarget_estimation_amd -Wl,-rpath,$PWD/target_estimation_amd/lib -Wl,-rpath,/opt/rocm/lib */
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include "target_manager_c.h"

static double now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }

int main(int argc, char** argv) {
  const char* file = argc > 1 ? argv[1] : "models/model_angular_velocities_params.yaml";
  const int dflt[] = {1, 3, 40, 400, 4000};
  const int ns = argc > 2 ? argc - 2 : 5;        /* sizes on the command line after the model file, else the reference's scale */
  for (int si = 0; si < ns; ++si) {
    const int T = argc > 2 ? atoi(argv[2 + si]) : dflt[si];
    if (T <= 0) return 2;
    const double ti = now();
    target_manager_c* m = target_manager_new(file);
    if (!m) return 3;
    double p[7] = {0.1, 0.2, 0.3, 0, 0, 0, 1.0}, pose[7], twist[6];
    for (int i = 0; i < T; ++i) target_manager_init(m, (unsigned)i, 0.004, p, 0.0);
    int ticks = 4000 / T; if (ticks < 20) ticks = 20; if (ticks > 500) ticks = 500;
    if (T > 4000) { ticks = 400000 / T; if (ticks < 3) ticks = 3; printf("%5d targets: %.2f us per target_manager_init\n", T, (now() - ti) / T * 1e6); }
    for (int pattern = 0; pattern < 2; ++pattern) {
      for (int k = -3; k < ticks; ++k) {
        static double t0; if (k == 0) t0 = now();
        if (pattern == 0) {                       /* all updates, then all reads (a node tick) */
          for (int i = 0; i < T; ++i) target_manager_update_meas(m, (unsigned)i, 0.004, p);
          for (int i = 0; i < T; ++i) { target_manager_get_est_pose(m, (unsigned)i, pose); target_manager_get_est_twist(m, (unsigned)i, twist); }
        } else {                                   /* update and read back target by target (the reference's test loop) */
          for (int i = 0; i < T; ++i) { target_manager_update_meas(m, (unsigned)i, 0.004, p); target_manager_get_est_pose(m, (unsigned)i, pose); target_manager_get_est_twist(m, (unsigned)i, twist); }
        }
        if (k == ticks - 1) {
          const double el = now() - t0;
          printf("%5d targets, %s: %9.1f us per tick, %7.2f us per target-cycle\n", T, pattern == 0 ? "batched reads    " : "interleaved reads", el / ticks * 1e6, el / ticks / T * 1e6);
          fflush(stdout);
        }
      }
    }
    target_manager_delete(m);
  }
  return 0;
}
