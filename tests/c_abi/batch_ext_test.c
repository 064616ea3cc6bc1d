/* batch_ext_test.c -- a plain C caller (gcc, C99, no HIP headers) of the batched extension in
 * include/target_estimation_amd/target_batch_c.h: per-class parameters, by-id calls with the ids in random order (resolved
 * on the device from 8192 ids per call), unknown ids, the by-id getters, the sphere query, and the pose gather on a
 * one-rank communicator.  The arithmetic is not re-derived here (the Python parity tests hold it to the oracle); this
 * program checks the C calling convention: argument order, array shapes, return values, that rows come back in the
 * caller's order, and that two classes of the same batch filter differently.
 * usage: batch_ext_test <uniform_acceleration model.yaml>   (exit code 0 = pass) */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "target_batch_c.h"

#define N 12000
#define NS 9   /* uniform_acceleration: 9 states, 3 measurements */
#define NM 3

static unsigned lcg(unsigned* s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s model.yaml\n", argv[0]); return 2; }
  target_manager_c* m = target_manager_new_ex(NULL, TARGET_DTYPE_F64, 0);
  if (!m) { fprintf(stderr, "target_manager_new_ex failed: %s\n", target_manager_last_error()); return 3; }
  int fail = 0;
  /* two parameter classes: the same Q, R scaled by 1 and by 100; diagonal matrices of the shipped kind */
  static double Q[2][NS * NS], R[2][NM * NM], P0[2][NS * NS];
  memset(Q, 0, sizeof Q); memset(R, 0, sizeof R); memset(P0, 0, sizeof P0);
  for (int c = 0; c < 2; ++c) {
    const double s = c ? 100.0 : 1.0;
    for (int i = 0; i < NS; ++i) { Q[c][i * NS + i] = 1e-6 * s; P0[c][i * NS + i] = 0.1; }
    for (int i = 0; i < NM; ++i) R[c][i * NM + i] = 1e-4 * s;
  }
  unsigned* ids = malloc(sizeof(unsigned) * N);
  unsigned* cls = malloc(sizeof(unsigned) * N);
  double* p0 = calloc((size_t)N * 7, sizeof(double));
  for (int i = 0; i < N; ++i) { ids[i] = 100000u + 3u * (unsigned)i; cls[i] = (unsigned)(i & 1); p0[i * 7 + 6] = 1.0; p0[i * 7] = (double)i * 1e-3; }
  const double dt = 0.004;
  long made = target_manager_init_batch_classes(m, TARGET_UNIFORM_ACCELERATION, ids, N, dt, 0.0, 2, &Q[0][0], &R[0][0], &P0[0][0], cls, p0, NULL, NULL);
  if (made != N) { fprintf(stderr, "init_batch_classes created %ld\n", made); fail = 1; }
  if (target_manager_num_batches(m) != 1) { fprintf(stderr, "classes must share one batch\n"); fail = 1; }
  target_batch_c* b = target_manager_get_batch(m, 0);
  if (!b || target_batch_num_classes(b) != 2 || target_batch_size(b) != N) { fprintf(stderr, "batch shape\n"); fail = 1; }

  /* by-id update in RANDOM order plus 5 unknown ids: >= 8192 entries -> resolved on the device */
  const int NQ = N + 5;
  unsigned* qid = malloc(sizeof(unsigned) * NQ);
  double* meas = calloc((size_t)NQ * 7, sizeof(double));
  int* perm = malloc(sizeof(int) * N);
  unsigned seed = 7u;
  for (int i = 0; i < N; ++i) perm[i] = i;
  for (int i = N - 1; i > 0; --i) { int j = (int)(lcg(&seed) % (unsigned)(i + 1)); int t = perm[i]; perm[i] = perm[j]; perm[j] = t; }
  for (int tick = 0; tick < 3; ++tick) {
    for (int k = 0; k < N; ++k) {
      const int i = perm[k];
      qid[k] = ids[i];
      meas[k * 7 + 0] = p0[i * 7] + 1.0;      /* a one-metre jump in x: the two classes must follow it differently */
      meas[k * 7 + 6] = 1.0;
    }
    for (int k = N; k < NQ; ++k) { qid[k] = 7u + (unsigned)k; meas[k * 7 + 6] = 1.0; }
    long stepped = target_manager_update_meas_batch(m, qid, NQ, dt, meas, NULL);
    if (stepped != N) { fprintf(stderr, "update_meas_batch stepped %ld of %d known ids\n", stepped, N); fail = 1; }
  }
  /* getters by id, same random order: rows in the caller's order, unknown ids flagged and left untouched */
  double* pose = malloc(sizeof(double) * (size_t)NQ * 7);
  unsigned char* found = malloc((size_t)NQ);
  for (int k = 0; k < NQ * 7; ++k) pose[k] = -777.0;
  long got = target_manager_get_est_batch(m, qid, NQ, pose, NULL, NULL, found);
  if (got != N) { fprintf(stderr, "get_est_batch found %ld\n", got); fail = 1; }
  double gain_sum[2] = {0, 0};
  for (int k = 0; k < N; ++k) {
    const int i = perm[k];
    if (!found[k]) { fail = 1; break; }
    const double moved = pose[k * 7] - p0[i * 7];              /* how far the estimate followed the jump */
    if (!(moved > 0.0 && moved < 1.0 + 1e-9)) { fprintf(stderr, "row %d (id %u): moved %g\n", k, qid[k], moved); fail = 1; break; }
    gain_sum[cls[i]] += moved;
  }
  for (int k = N; k < NQ; ++k)
    if (found[k] || pose[k * 7] != -777.0) { fprintf(stderr, "unknown id row %d was touched\n", k); fail = 1; }
  if (!(fabs(gain_sum[0] - gain_sum[1]) > 1e-3 * N)) { fprintf(stderr, "the two classes filtered alike: %g %g\n", gain_sum[0], gain_sum[1]); fail = 1; }
  int n_meas = target_manager_get_n_measurements(m, ids[5]);
  if (n_meas != 3) { fprintf(stderr, "n_measurements %d\n", n_meas); fail = 1; }

  /* sphere query by id (host arrays) */
  double origin[3] = {0, 0, 0};
  double* delta = malloc(sizeof(double) * N);
  long asked = target_manager_intersect_sphere_batch(m, ids, N, 3 * dt, origin, 1000.0, delta, NULL, found);
  if (asked != N) { fprintf(stderr, "intersect_sphere_batch %ld\n", asked); fail = 1; }

  /* pose gather on a one-rank communicator: the root's own rows, batch order then slot order (= ids order here) */
  char id128[128];
  if (target_comm_unique_id(id128) != 0) { fprintf(stderr, "target_comm_unique_id: %s\n", target_manager_last_error()); fail = 1; }
  else {
    target_comm_c* comm = target_comm_new(id128, 0, 1);
    if (!comm) { fprintf(stderr, "target_comm_new: %s\n", target_manager_last_error()); fail = 1; }
    else {
      long counts[1] = {N};
      /* a device buffer without HIP headers: borrow the batch's own getter output path -- the gather needs a device pointer,
       * so this C program only checks the argument validation (a NULL receive buffer on the root is refused) */
      if (target_manager_gather_pose_begin(m, comm, 0, counts, NULL) == 0) { fprintf(stderr, "NULL receive buffer accepted\n"); fail = 1; }
      long wrong[1] = {N - 1};
      if (target_manager_gather_pose_begin(m, comm, 0, wrong, NULL) == 0) { fprintf(stderr, "wrong counts accepted\n"); fail = 1; }
      float ms = -1.f;
      if (target_manager_gather_pose_wait(comm, &ms) != 0) { fprintf(stderr, "gather wait\n"); fail = 1; }
      target_comm_delete(comm);
    }
  }
  long erased = target_manager_erase_batch(m, ids, N / 2);
  if (erased != N / 2 || target_manager_size(m) != N - N / 2) { fprintf(stderr, "erase_batch\n"); fail = 1; }
  long again = target_manager_update_meas_batch(m, qid, NQ, dt, NULL, NULL);   /* predict-only by id after the erase */
  if (again != N - N / 2) { fprintf(stderr, "after erase: stepped %ld\n", again); fail = 1; }
  target_manager_delete(m);
  printf(fail ? "BATCH EXT TEST FAILED\n" : "batch extension test ok\n");
  return fail;
}
