/*
 * te_oracle_impl.h -- body of the CPU oracle, instantiated once per precision by
 * te_oracle.c (REAL = double / float, SFX = f64 / f32).  TEST INFRASTRUCTURE ONLY,
 * see te_oracle.h for the parity status and the rules on who may use it.
 *
 * All citations are file:line under /root/reference.
 */

#ifndef REAL
#error "include from te_oracle.c"
#endif

#define ORC_CAT_(a, b) a##_##b
#define ORC_CAT(a, b) ORC_CAT_(a, b)
#define FN(name) ORC_CAT(name, SFX)
#define TGT FN(orc_target)

#define R_PI ((REAL)M_PI)

/* ------------------------------------------------------------------------- */
/* small dense helpers: out = A(r x k) * B(k x c), k-ascending accumulation    */
/* ------------------------------------------------------------------------- */
static void FN(mm)(REAL* out, const REAL* A, const REAL* B, int r, int k, int c) {
  for (int i = 0; i < r; ++i)
    for (int j = 0; j < c; ++j) {
      REAL acc = A[i * k] * B[j];
      for (int l = 1; l < k; ++l) acc += A[i * k + l] * B[l * c + j];
      out[i * c + j] = acc;
    }
}

/* out = A(r x k) * B^T, B is (c x k) */
static void FN(mmt)(REAL* out, const REAL* A, const REAL* B, int r, int k, int c) {
  for (int i = 0; i < r; ++i)
    for (int j = 0; j < c; ++j) {
      REAL acc = A[i * k] * B[j * k];
      for (int l = 1; l < k; ++l) acc += A[i * k + l] * B[j * k + l];
      out[i * c + j] = acc;
    }
}

/* Partial-pivot LU inverse: what Eigen's MatrixXd::inverse() does for dynamic sizes
 * (PartialPivLU, unblocked for small matrices; then solve against the identity).
 * Call site: src/kalman.cpp:92,137.  Returns 0 on success, -1 on an exactly zero pivot. */
int FN(orc_inverse)(int n, const REAL* Ain, REAL* Ainv) {
  REAL lu[ORC_NMAX * ORC_NMAX];
  int perm[ORC_NMAX];
  int rc = 0;
  for (int i = 0; i < n * n; ++i) lu[i] = Ain[i];
  for (int i = 0; i < n; ++i) perm[i] = i;
  for (int k = 0; k < n; ++k) {
    int piv = k;
    REAL best = RFABS(lu[k * n + k]);
    for (int i = k + 1; i < n; ++i) {
      REAL v = RFABS(lu[i * n + k]);
      if (v > best) { best = v; piv = i; }
    }
    if (best == (REAL)0) { rc = -1; continue; }
    if (piv != k) {
      for (int j = 0; j < n; ++j) { REAL t = lu[k * n + j]; lu[k * n + j] = lu[piv * n + j]; lu[piv * n + j] = t; }
      int t = perm[k]; perm[k] = perm[piv]; perm[piv] = t;
    }
    for (int i = k + 1; i < n; ++i) lu[i * n + k] /= lu[k * n + k];
    for (int i = k + 1; i < n; ++i)
      for (int j = k + 1; j < n; ++j) lu[i * n + j] -= lu[i * n + k] * lu[k * n + j];
  }
  /* solve L U X = P I, column by column */
  for (int c = 0; c < n; ++c) {
    REAL y[ORC_NMAX];
    for (int i = 0; i < n; ++i) y[i] = (perm[i] == c) ? (REAL)1 : (REAL)0;
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < i; ++j) y[i] -= lu[i * n + j] * y[j];
    for (int i = n - 1; i >= 0; --i) {
      for (int j = i + 1; j < n; ++j) y[i] -= lu[i * n + j] * y[j];
      y[i] /= lu[i * n + i];
    }
    for (int i = 0; i < n; ++i) Ainv[i * n + c] = y[i];
  }
  return rc;
}

/* ------------------------------------------------------------------------- */
/* geometry hot subset: include/target_estimation/geometry.hpp                 */
/* ------------------------------------------------------------------------- */

/* geometry.hpp:31-36 */
REAL FN(orc_constrain_angle)(REAL x) {
  x = RFMOD(x + R_PI, 2 * R_PI);
  if (x < 0) x += 2 * R_PI;
  return x - R_PI;
}

/* geometry.hpp:43-45 */
REAL FN(orc_angle_conv)(REAL angle) { return RFMOD(FN(orc_constrain_angle)(angle), 2 * R_PI); }

/* geometry.hpp:53-58 */
REAL FN(orc_angle_diff)(REAL a, REAL b) {
  REAL dif = RFMOD(b - a + R_PI, 2 * R_PI);
  if (dif < 0) dif += 2 * R_PI;
  return dif - R_PI;
}

/* geometry.hpp:66-68 and :70-76 (per component) */
REAL FN(orc_unwrap)(REAL previousAngle, REAL newAngle) {
  return previousAngle - FN(orc_angle_diff)(newAngle, FN(orc_angle_conv)(previousAngle));
}

/* Eigen::Quaterniond::normalize(): coeffs /= sqrt(squaredNorm).  q = [x y z w]. */
void FN(orc_quat_normalize)(REAL* q) {
  REAL nrm = RSQRT(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  q[0] /= nrm; q[1] /= nrm; q[2] /= nrm; q[3] /= nrm;
}

/* geometry.hpp:154-176.  q = [x y z w], rpy = [roll pitch yaw] */
void FN(orc_quat_to_rpy)(const REAL* q, REAL* rpy) {
  const REAL x = q[0], y = q[1], z = q[2], w = q[3];
  if (-2 * (x * z - w * y) > (REAL)0.9999) {
    rpy[0] = 0;
    rpy[1] = R_PI / 2;
    rpy[2] = 2 * RATAN2(z, w);
  } else if (-2 * (x * z - w * y) < (REAL)-0.9999) {
    rpy[0] = 0;
    rpy[1] = -R_PI / 2;
    rpy[2] = 2 * RATAN2(z, w);
  } else {
    rpy[0] = RATAN2(2 * (y * z + w * x), (w * w - x * x - y * y + z * z));
    rpy[1] = RASIN(-2 * (x * z - w * y));
    rpy[2] = RATAN2(2 * (x * y + w * z), (w * w + x * x - y * y - z * z));
  }
}

/* geometry.hpp:178-189 */
void FN(orc_rpy_to_quat)(const REAL* rpy, REAL* q) {
  REAL phi = rpy[0] / 2, the = rpy[1] / 2, psi = rpy[2] / 2;
  q[3] = RCOS(phi) * RCOS(the) * RCOS(psi) + RSIN(phi) * RSIN(the) * RSIN(psi);
  q[0] = RSIN(phi) * RCOS(the) * RCOS(psi) - RCOS(phi) * RSIN(the) * RSIN(psi);
  q[1] = RCOS(phi) * RSIN(the) * RCOS(psi) + RSIN(phi) * RCOS(the) * RSIN(psi);
  q[2] = RCOS(phi) * RCOS(the) * RSIN(psi) - RSIN(phi) * RSIN(the) * RCOS(psi);
  FN(orc_quat_normalize)(q);
}

/* geometry.hpp:191-196.  R row-major 3x3 */
void FN(orc_rot_to_rpy)(const REAL* R, REAL* rpy) {
  rpy[0] = RATAN2(R[7], R[8]);
  rpy[1] = RATAN2(-R[6], RSQRT(R[7] * R[7] + R[8] * R[8]));
  rpy[2] = RATAN2(R[3], R[0]);
}

/* Eigen::Quaterniond::toRotationMatrix() (Eigen/src/Geometry/Quaternion.h), used at
 * angular_rates.cpp:127 and angular_velocities.cpp:163 */
void FN(orc_quat_to_rot)(const REAL* q, REAL* R) {
  const REAL x = q[0], y = q[1], z = q[2], w = q[3];
  const REAL tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const REAL twx = tx * w, twy = ty * w, twz = tz * w;
  const REAL txx = tx * x, txy = ty * x, txz = tz * x;
  const REAL tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

/* Eigen's rotation-matrix -> quaternion assignment (Quaternion.h, quaternionbase_assign_impl
 * <Other,3,3>), used by isometryToPose7d at geometry.hpp:593 */
void FN(orc_rot_to_quat)(const REAL* R, REAL* q) {
  REAL t = R[0] + R[4] + R[8];
  if (t > 0) {
    t = RSQRT(t + 1);
    q[3] = (REAL)0.5 * t;
    t = (REAL)0.5 / t;
    q[0] = (R[7] - R[5]) * t;
    q[1] = (R[2] - R[6]) * t;
    q[2] = (R[3] - R[1]) * t;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[i * 3 + i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = RSQRT(R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k] + 1);
    q[i] = (REAL)0.5 * t;
    t = (REAL)0.5 / t;
    q[3] = (R[k * 3 + j] - R[j * 3 + k]) * t;
    q[j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
    q[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
  }
}

/* geometry.hpp:333-351 */
void FN(orc_rpy_to_ear_base)(const REAL* rpy, REAL* E) {
  REAL c_r = RCOS(rpy[0]), s_r = RSIN(rpy[0]), c_p = RCOS(rpy[1]), s_p = RSIN(rpy[1]);
  E[0] = 1; E[1] = 0;    E[2] = -s_p;
  E[3] = 0; E[4] = c_r;  E[5] = c_p * s_r;
  E[6] = 0; E[7] = -s_r; E[8] = c_p * c_r;
}

/* geometry.hpp:359-374 */
void FN(orc_rpy_to_ear_base_inv)(const REAL* rpy, REAL* E) {
  REAL c_r = RCOS(rpy[0]), s_r = RSIN(rpy[0]), c_p = RCOS(rpy[1]), s_p = RSIN(rpy[1]);
  E[0] = 1; E[1] = (s_p * s_r) / c_p; E[2] = (c_r * s_p) / c_p;
  E[3] = 0; E[4] = c_r;               E[5] = -s_r;
  E[6] = 0; E[7] = s_r / c_p;         E[8] = c_r / c_p;
}

/* geometry.hpp:394-410 */
void FN(orc_ear_base_inv_jac_rpy)(const REAL* rpy, const REAL* omega, REAL dt, REAL* J) {
  REAL wy = omega[1], wz = omega[2];
  REAL c_r = RCOS(rpy[0]), c_p = RCOS(rpy[1]), s_r = RSIN(rpy[0]), s_p = RSIN(rpy[1]);
  J[0] = (dt * (wy * c_r * s_p - wz * s_p * s_r)) / c_p + 1;
  J[1] = (dt * (wz * c_r + wy * s_r)) / (c_p * c_p);
  J[2] = 0;
  J[3] = -dt * (wz * c_r + wy * s_r);
  J[4] = 1;
  J[5] = 0;
  J[6] = (dt * (wy * c_r - wz * s_r)) / c_p;
  J[7] = (dt * s_p * (wz * c_r + wy * s_r)) / (c_p * c_p);
  J[8] = 1;
}

/* geometry.hpp:412-426 */
void FN(orc_ear_base_inv_jac_omega)(const REAL* rpy, REAL dt, REAL* J) {
  REAL c_r = RCOS(rpy[0]), c_p = RCOS(rpy[1]), s_r = RSIN(rpy[0]), s_p = RSIN(rpy[1]);
  J[0] = dt; J[1] = (dt * s_p * s_r) / c_p; J[2] = (dt * c_r * s_p) / c_p;
  J[3] = 0;  J[4] = dt * c_r;               J[5] = -dt * s_r;
  J[6] = 0;  J[7] = (dt * s_r) / c_p;       J[8] = (dt * c_r) / c_p;
}

/* geometry.hpp:448-465 (omegaToMatrix) and :493-504 (Qtran); M row-major 4x4, q = [x y z w] */
void FN(orc_qtran)(REAL dt, const REAL* omega, REAL* M) {
  REAL omega_norm = RSQRT(omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2]);
  REAL tmp = omega_norm * dt / (REAL)2;
  REAL S[16] = {0, -omega[2], omega[1], omega[0],
                omega[2], 0, -omega[0], omega[1],
                -omega[1], omega[0], 0, omega[2],
                -omega[0], -omega[1], -omega[2], 0};
  for (int i = 0; i < 16; ++i) S[i] = (REAL)0.5 * S[i];
  if (omega_norm > 0) {
    REAL c = RCOS(tmp), s = (REAL)2 / omega_norm * RSIN(tmp);
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) M[i * 4 + j] = c * (i == j ? (REAL)1 : (REAL)0) + s * S[i * 4 + j];
  } else {
    for (int i = 0; i < 16; ++i) M[i] = (i % 5 == 0) ? (REAL)1 : (REAL)0;
  }
}

/* geometry.hpp:619-628 */
static void FN(pose7_to_pose6)(const REAL* p7, REAL* p6) {
  REAL q[4] = {p7[3], p7[4], p7[5], p7[6]};
  p6[0] = p7[0]; p6[1] = p7[1]; p6[2] = p7[2];
  FN(orc_quat_normalize)(q);
  FN(orc_quat_to_rpy)(q, p6 + 3);
}

/* geometry.hpp:602-608; Isometry3d::rotation() == linear() (Eigen >= 3.3) */
static void FN(isometry_to_pose6)(const TGT* tg, REAL* p6) {
  p6[0] = tg->T_trans[0]; p6[1] = tg->T_trans[1]; p6[2] = tg->T_trans[2];
  FN(orc_rot_to_rpy)(tg->T_lin, p6 + 3);
}

/* ------------------------------------------------------------------------- */
/* Kalman estimators: src/kalman.cpp                                          */
/* ------------------------------------------------------------------------- */

/* TargetAngularVelocities::f, src/types/angular_velocities.cpp:126-140 */
static void FN(av_f)(const REAL* x, REAL dt, REAL* out) {
  REAL Einv[9];
  for (int i = 0; i < 12; ++i) out[i] = 0;
  FN(orc_rpy_to_ear_base_inv)(x + 3, Einv);
  for (int i = 0; i < 3; ++i) out[i] = x[i] + dt * x[6 + i];
  for (int i = 0; i < 3; ++i) out[6 + i] = x[6 + i];
  for (int i = 0; i < 3; ++i) out[9 + i] = x[9 + i];
  for (int i = 0; i < 3; ++i) {
    REAL acc = (dt * Einv[i * 3]) * x[9];
    acc += (dt * Einv[i * 3 + 1]) * x[10];
    acc += (dt * Einv[i * 3 + 2]) * x[11];
    out[3 + i] = x[3 + i] + acc;
  }
}

/* LinearKalmanFilter::predict src/kalman.cpp:84-88; ExtendedKalmanFilter::predict :129-133 */
static void FN(kf_predict)(TGT* tg) {
  const int n = tg->n;
  REAL AP[ORC_NMAX * ORC_NMAX], APAt[ORC_NMAX * ORC_NMAX];
  if (tg->model == ORC_ANGULAR_VELOCITIES) {
    FN(av_f)(tg->x_hat, tg->f_dt, tg->x_hat_new);
  } else {
    FN(mm)(tg->x_hat_new, tg->A, tg->x_hat, n, n, 1);
  }
  FN(mm)(AP, tg->A, tg->P, n, n, n);
  FN(mmt)(APAt, AP, tg->A, n, n, n);
  for (int i = 0; i < n * n; ++i) tg->P[i] = APAt[i] + tg->Q[i];
}

/* LinearKalmanFilter::estimate src/kalman.cpp:90-95; ExtendedKalmanFilter::estimate :135-140
 * (h(x) = x[0:6] = C*x for the only EKF model, angular_velocities.cpp:142-151) */
static void FN(kf_estimate)(TGT* tg, const REAL* y) {
  const int n = tg->n, m = tg->m;
  REAL PCt[ORC_NMAX * ORC_MMAX], CP[ORC_MMAX * ORC_NMAX], S[ORC_MMAX * ORC_MMAX] = {0}, Sinv[ORC_MMAX * ORC_MMAX];
  REAL Cx[ORC_MMAX], innov[ORC_MMAX], Kin[ORC_NMAX], KC[ORC_NMAX * ORC_NMAX], IKC[ORC_NMAX * ORC_NMAX];
  REAL Pn[ORC_NMAX * ORC_NMAX];
  FN(mmt)(PCt, tg->P, tg->C, n, n, m);
  FN(mm)(CP, tg->C, tg->P, m, n, n);
  FN(mmt)(S, CP, tg->C, m, n, m);
  for (int i = 0; i < m * m; ++i) S[i] += tg->R[i];
  FN(orc_inverse)(m, S, Sinv);
  FN(mm)(tg->K, PCt, Sinv, n, m, m);
  if (tg->model == ORC_ANGULAR_VELOCITIES) {
    for (int i = 0; i < m; ++i) Cx[i] = tg->x_hat_new[i]; /* h_, angular_velocities.cpp:146-148 */
  } else {
    FN(mm)(Cx, tg->C, tg->x_hat_new, m, n, 1);
  }
  for (int i = 0; i < m; ++i) innov[i] = y[i] - Cx[i];
  FN(mm)(Kin, tg->K, innov, n, m, 1);
  for (int i = 0; i < n; ++i) tg->x_hat_new[i] += Kin[i];
  FN(mm)(KC, tg->K, tg->C, n, m, n);
  for (int i = 0; i < n * n; ++i) IKC[i] = tg->Id[i] - KC[i];
  FN(mm)(Pn, IKC, tg->P, n, n, n);
  for (int i = 0; i < n * n; ++i) tg->P[i] = Pn[i];
}

/* KalmanFilterInterface::update(y) src/kalman.cpp:30-42, update() :44-54 */
static void FN(kf_update)(TGT* tg, const REAL* y) {
  FN(kf_predict)(tg);
  if (y) FN(kf_estimate)(tg, y);
  for (int i = 0; i < tg->n; ++i) tg->x_hat[i] = tg->x_hat_new[i];
}

/* ------------------------------------------------------------------------- */
/* per-model target logic: src/types (all four)                                   */
/* ------------------------------------------------------------------------- */

/* uniform_velocity.cpp:90-96, uniform_acceleration.cpp:91-99, angular_rates.cpp:108-115,
 * angular_velocities.cpp:116-124 (evaluated at the target's copy of the posterior, x_) */
static void FN(update_A)(TGT* tg, REAL dt) {
  const int n = tg->n;
  REAL* A = tg->A;
  switch (tg->model) {
    case ORC_UNIFORM_VELOCITY:
      for (int i = 0; i < n; ++i) A[i * n + i] = 1;
      for (int i = 0; i < n / 2; ++i) A[i * n + i + n / 2] = (REAL)1 * dt;
      break;
    case ORC_UNIFORM_ACCELERATION:
      for (int i = 0; i < n * n; ++i) A[i] = 0;
      /* fall through */
    case ORC_ANGULAR_RATES:
      for (int i = 0; i < n; ++i) A[i * n + i] = 1;
      for (int i = 0; i < (n * 2) / 3; ++i) A[i * n + i + n / 3] = (REAL)1 * dt;
      for (int i = 0; i < n / 3; ++i) A[i * n + i + (n * 2) / 3] = (REAL)1 * (REAL)0.5 * dt * dt;
      break;
    case ORC_ANGULAR_VELOCITIES: {
      REAL Jr[9], Jw[9];
      FN(orc_ear_base_inv_jac_rpy)(tg->x + 3, tg->x + 9, dt, Jr);
      FN(orc_ear_base_inv_jac_omega)(tg->x + 3, dt, Jw);
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
          REAL d = (i == j) ? (REAL)1 : (REAL)0;
          A[i * n + j] = d;
          A[i * n + 6 + j] = d * dt;
          A[(3 + i) * n + 3 + j] = Jr[i * 3 + j];
          A[(3 + i) * n + 9 + j] = Jw[i * 3 + j];
          A[(6 + i) * n + 6 + j] = d;
          A[(9 + i) * n + 9 + j] = d;
        }
    } break;
  }
}

/* uniform_velocity.cpp:98-115, uniform_acceleration.cpp:101-118, angular_rates.cpp:117-138,
 * angular_velocities.cpp:153-169 */
static void FN(update_target_state)(TGT* tg) {
  const int n = tg->n;
  for (int i = 0; i < n; ++i) tg->x[i] = tg->x_hat[i];
  for (int i = 0; i < n * n; ++i) tg->P_tgt[i] = tg->P[i];
  tg->T_trans[0] = tg->x[0]; tg->T_trans[1] = tg->x[1]; tg->T_trans[2] = tg->x[2];
  switch (tg->model) {
    case ORC_UNIFORM_VELOCITY:
    case ORC_UNIFORM_ACCELERATION:
      for (int i = 0; i < 9; ++i) tg->T_lin[i] = (i % 4 == 0) ? (REAL)1 : (REAL)0;
      for (int i = 0; i < 3; ++i) { tg->twist[i] = tg->x[3 + i]; tg->twist[3 + i] = 0; }
      for (int i = 0; i < 3; ++i) {
        tg->acceleration[i] = (tg->model == ORC_UNIFORM_ACCELERATION) ? tg->x[6 + i] : (REAL)0;
        tg->acceleration[3 + i] = 0;
      }
      break;
    case ORC_ANGULAR_RATES: {
      REAL q[4], rpy[3], Ear[9];
      FN(orc_rpy_to_quat)(tg->x + 3, q);
      FN(orc_quat_to_rot)(q, tg->T_lin);
      for (int i = 0; i < 3; ++i) tg->twist[i] = tg->x[6 + i];
      FN(orc_rot_to_rpy)(tg->T_lin, rpy);
      FN(orc_rpy_to_ear_base)(rpy, Ear);
      FN(mm)(tg->twist + 3, Ear, tg->x + 9, 3, 3, 1);
      for (int i = 0; i < 6; ++i) tg->acceleration[i] = tg->x[12 + i];
    } break;
    case ORC_ANGULAR_VELOCITIES: {
      REAL q[4];
      FN(orc_rpy_to_quat)(tg->x + 3, q);
      FN(orc_quat_to_rot)(q, tg->T_lin);
      for (int i = 0; i < 6; ++i) tg->twist[i] = tg->x[6 + i];
      /* acceleration_ stays at the base ctor's zeros (target_interface.cpp:27) */
    } break;
  }
  FN(isometry_to_pose6)(tg, tg->pose_internal);
}

int FN(orc_target_sizeof)(void) { return (int)sizeof(TGT); }

/* TargetInterface ctor src/target_interface.cpp:18-41 + model ctors
 * (uniform_velocity.cpp:16-61, uniform_acceleration.cpp:17-62, angular_rates.cpp:21-70,
 * angular_velocities.cpp:21-78) + LinearKalmanFilter ctor src/kalman.cpp:62-82 + init :16-21.
 * Q,R,P0 row-major n*n / m*m doubles (symmetric in every shipped model, so the reference's
 * column-major Map of the row-major YAML list, target_manager.cpp:25, is immaterial).
 * Deliberate choice: meas_rpy_internal_ (never initialised in the reference,
 * angular_rates.hpp:110 / angular_velocities.hpp:127) is ZERO here. */
int FN(orc_target_init)(TGT* tg, int model, unsigned id, double dt0, double t0, const double* Q,
                        const double* R, const double* P0, const double* p0, const double* v0,
                        const double* a0) {
  static const double zero6[6] = {0, 0, 0, 0, 0, 0};
  int n, m;
  switch (model) {
    case ORC_UNIFORM_VELOCITY: n = 6; m = 3; break;
    case ORC_UNIFORM_ACCELERATION: n = 9; m = 3; break;
    case ORC_ANGULAR_RATES: n = 18; m = 6; break;
    case ORC_ANGULAR_VELOCITIES: n = 12; m = 6; break;
    default: return -1;
  }
  if (!v0) v0 = zero6;
  if (!a0) a0 = zero6;
  memset(tg, 0, sizeof(*tg));
  tg->model = model; tg->n = n; tg->m = m; tg->id = id;
  tg->n_meas = 0; tg->t = t0;
  for (int i = 0; i < 9; ++i) tg->T_lin[i] = (i % 4 == 0) ? (REAL)1 : (REAL)0;
  tg->measured_pose[6] = 1;
  for (int i = 0; i < n * n; ++i) { tg->Q[i] = (REAL)Q[i]; tg->P0[i] = (REAL)P0[i]; tg->P_tgt[i] = (REAL)P0[i]; }
  for (int i = 0; i < m * m; ++i) tg->R[i] = (REAL)R[i];
  for (int i = 0; i < n; ++i) tg->Id[i * n + i] = 1;
  for (int i = 0; i < m; ++i) tg->C[i * n + i] = 1;

  REAL p7[7];
  for (int i = 0; i < 7; ++i) p7[i] = (REAL)p0[i];
  switch (model) {
    case ORC_UNIFORM_VELOCITY:
      for (int i = 0; i < 3; ++i) { tg->x[i] = p7[i]; tg->x[3 + i] = (REAL)v0[i]; }
      break;
    case ORC_UNIFORM_ACCELERATION:
      for (int i = 0; i < 3; ++i) { tg->x[i] = p7[i]; tg->x[3 + i] = (REAL)v0[i]; tg->x[6 + i] = (REAL)a0[i]; }
      break;
    case ORC_ANGULAR_RATES:
      FN(pose7_to_pose6)(p7, tg->pose_internal);
      for (int i = 0; i < 6; ++i) { tg->x[i] = tg->pose_internal[i]; tg->x[6 + i] = (REAL)v0[i]; tg->x[12 + i] = (REAL)a0[i]; }
      break;
    case ORC_ANGULAR_VELOCITIES:
      FN(pose7_to_pose6)(p7, tg->pose_internal);
      for (int i = 0; i < 6; ++i) { tg->x[i] = tg->pose_internal[i]; tg->x[6 + i] = (REAL)v0[i]; }
      break;
  }
  FN(update_A)(tg, (REAL)dt0);
  tg->f_dt = (REAL)dt0;
  /* estimator_->init(x_): src/kalman.cpp:16-21 */
  for (int i = 0; i < n; ++i) tg->x_hat[i] = tg->x[i];
  for (int i = 0; i < n * n; ++i) tg->P[i] = tg->P0[i];
  tg->initialized = 1;
  FN(update_target_state)(tg);
  return 0;
}

/* addMeasurement: uniform_velocity.cpp:63-76, uniform_acceleration.cpp:64-77,
 * angular_rates.cpp:72-94, angular_velocities.cpp:80-102 */
void FN(orc_target_add_measurement)(TGT* tg, double dt_d, const double* meas) {
  const REAL dt = (REAL)dt_d;
  REAL y[ORC_MMAX];
  FN(update_A)(tg, dt);
  /* updateMeasurement, target_interface.cpp:142-146 */
  for (int i = 0; i < 7; ++i) tg->measured_pose[i] = (REAL)meas[i];
  tg->n_meas += 1;
  y[0] = tg->measured_pose[0]; y[1] = tg->measured_pose[1]; y[2] = tg->measured_pose[2];
  if (tg->model == ORC_ANGULAR_RATES || tg->model == ORC_ANGULAR_VELOCITIES) {
    REAL q[4] = {tg->measured_pose[3], tg->measured_pose[4], tg->measured_pose[5], tg->measured_pose[6]};
    REAL rpy[3];
    FN(orc_quat_normalize)(q);
    FN(orc_quat_to_rpy)(q, rpy);
    for (int i = 0; i < 3; ++i) {
      rpy[i] = FN(orc_unwrap)(tg->meas_rpy_internal[i], rpy[i]);
      y[3 + i] = rpy[i];
      tg->meas_rpy_internal[i] = rpy[i];
    }
  }
  tg->f_dt = dt;
  FN(kf_update)(tg, y);
  FN(update_target_state)(tg);
  tg->t = tg->t + dt_d; /* updateTime, target_interface.cpp:148-152 */
}

/* update(dt) predict-only: uniform_velocity.cpp:78-88 and the same lines in the other models */
void FN(orc_target_update)(TGT* tg, double dt_d) {
  const REAL dt = (REAL)dt_d;
  FN(update_A)(tg, dt);
  tg->f_dt = dt;
  FN(kf_update)(tg, (const REAL*)0);
  FN(update_target_state)(tg);
  tg->t = tg->t + dt_d;
}

void FN(orc_target_get_state)(const TGT* tg, double* x, double* P) {
  if (x) for (int i = 0; i < tg->n; ++i) x[i] = (double)tg->x[i];
  if (P) for (int i = 0; i < tg->n * tg->n; ++i) P[i] = (double)tg->P_tgt[i];
}

/* TargetInterface::getEstimatedPose() target_interface.cpp:100-104 -> isometryToPose7d geometry.hpp:590-594 */
void FN(orc_target_get_pose)(const TGT* tg, double* pose7) {
  REAL q[4];
  FN(orc_rot_to_quat)(tg->T_lin, q);
  for (int i = 0; i < 3; ++i) pose7[i] = (double)tg->T_trans[i];
  for (int i = 0; i < 4; ++i) pose7[3 + i] = (double)q[i];
}

/* target_interface.cpp:106-109 */
void FN(orc_target_get_twist)(const TGT* tg, double* twist6) {
  for (int i = 0; i < 6; ++i) twist6[i] = (double)tg->twist[i];
}

/* target_interface.cpp:111-115 */
void FN(orc_target_get_acceleration)(const TGT* tg, double* acc6) {
  for (int i = 0; i < 6; ++i) acc6[i] = (double)tg->acceleration[i];
}

/* TargetInterface::getMeasuredPose target_interface.cpp:117-121 (measured_pose_: initPose, then the last measurement :142-146) */
void FN(orc_target_get_measured_pose)(const TGT* tg, double* pose7) {
  for (int i = 0; i < 7; ++i) pose7[i] = (double)tg->measured_pose[i];
}

/* pose_internal_ = [xyz rpy] (isometryToPose6d, geometry.hpp:602-608): the rt_logger "pose" channel, target_interface.cpp:36 */
void FN(orc_target_get_pose6)(const TGT* tg, double* pose6) {
  for (int i = 0; i < 6; ++i) pose6[i] = (double)tg->pose_internal[i];
}

/* TargetInterface::getEstimatedTransform target_interface.cpp:95-98: T_ as a row-major 4x4 */
void FN(orc_target_get_transform)(const TGT* tg, double* T16) {
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) T16[r * 4 + c] = (double)tg->T_lin[r * 3 + c];
    T16[r * 4 + 3] = (double)tg->T_trans[r];
  }
  T16[12] = 0; T16[13] = 0; T16[14] = 0; T16[15] = 1;
}

/* TargetInterface::getPeriodEstimate target_interface.cpp:80-87 */
double FN(orc_target_get_period_estimate)(const TGT* tg) {
  const double wx = (double)tg->twist[3], wy = (double)tg->twist[4], wz = (double)tg->twist[5];
  const double omega_norm = sqrt(wx * wx + wy * wy + wz * wz);
  if (omega_norm > 0) return 2 * M_PI / omega_norm;
  return -1.0;
}

/* getEstimatedPose(t1): uniform_velocity.cpp:117-127, uniform_acceleration.cpp:120-130,
 * angular_rates.cpp:140-151, angular_velocities.cpp:171-184 */
void FN(orc_target_get_pose_at)(const TGT* tg, double t1, double* pose7) {
  const REAL d = (REAL)(t1 - tg->t);
  REAL q[4] = {0, 0, 0, 1};
  REAL p[3] = {0, 0, 0};
  switch (tg->model) {
    case ORC_UNIFORM_VELOCITY:
      for (int i = 0; i < 3; ++i) p[i] = tg->T_trans[i] + tg->twist[i] * d;
      break;
    case ORC_UNIFORM_ACCELERATION:
      for (int i = 0; i < 3; ++i) p[i] = tg->T_trans[i] + tg->twist[i] * d + (REAL)0.5 * tg->acceleration[i] * d * d;
      break;
    case ORC_ANGULAR_RATES: {
      REAL v6[6];
      for (int i = 0; i < 6; ++i) v6[i] = tg->pose_internal[i] + tg->twist[i] * d + (REAL)0.5 * tg->acceleration[i] * d * d;
      FN(orc_rpy_to_quat)(v6 + 3, q);
      FN(orc_quat_normalize)(q);
      for (int i = 0; i < 3; ++i) p[i] = v6[i];
    } break;
    case ORC_ANGULAR_VELOCITIES: {
      REAL M[16], q0[4];
      for (int i = 0; i < 3; ++i) p[i] = tg->T_trans[i] + tg->twist[i] * d;
      FN(orc_rpy_to_quat)(tg->pose_internal + 3, q0);
      FN(orc_qtran)(d, tg->twist + 3, M);
      FN(mm)(q, M, q0, 4, 4, 1);
      FN(orc_quat_normalize)(q);
    } break;
  }
  for (int i = 0; i < 3; ++i) pose7[i] = (double)p[i];
  for (int i = 0; i < 4; ++i) pose7[3 + i] = (double)q[i];
}

/* getEstimatedTwist(t1): uniform_velocity.cpp:129-133, uniform_acceleration.cpp:132-136,
 * angular_rates.cpp:153-157; base target_interface.cpp:130-134 (angular_velocities) */
void FN(orc_target_get_twist_at)(const TGT* tg, double t1, double* twist6) {
  const REAL d = (REAL)(t1 - tg->t);
  for (int i = 0; i < 6; ++i) {
    REAL v = tg->twist[i];
    if (tg->model == ORC_UNIFORM_ACCELERATION || tg->model == ORC_ANGULAR_RATES) v = tg->twist[i] + tg->acceleration[i] * d;
    twist6[i] = (double)v;
  }
}

/* base getEstimatedAcceleration(t1), target_interface.cpp:136-140 (no model overrides it) */
void FN(orc_target_get_acceleration_at)(const TGT* tg, double t1, double* a6) {
  (void)t1;
  for (int i = 0; i < 6; ++i) a6[i] = (double)tg->acceleration[i];
}

/* IntersectionSolver::getIntersectionTimeWithSphere src/intersection_solver.cpp:42-89.
 * The reference solves in double whatever the filter precision; the quartic is built from the
 * (REAL) getters and solved in double. */
double FN(orc_intersection_time)(const TGT* tg, double t1, const double* origin, double radius) {
  double p7[7], tw[6], ac[6], coeff[5];
  FN(orc_target_get_pose_at)(tg, t1, p7);
  FN(orc_target_get_twist_at)(tg, t1, tw);
  FN(orc_target_get_acceleration_at)(tg, t1, ac);
  double x = p7[0] - origin[0], y = p7[1] - origin[1], z = p7[2] - origin[2];
  double vx = tw[0], vy = tw[1], vz = tw[2], ax = ac[0], ay = ac[1], az = ac[2], R = radius;
  coeff[4] = 0.25 * (ax * ax + ay * ay + az * az);
  coeff[3] = vx * ax + vy * ay + vz * az;
  coeff[2] = vx * vx + vy * vy + vz * vz + x * ax + y * ay + z * az;
  coeff[1] = 2 * (x * vx + y * vy + z * vz);
  coeff[0] = x * x + y * y + z * z - R * R;
  double d = orc_lowest_real_root(coeff, 5);
  if (d < 0) return -1;
  return d;
}

/* IntersectionSolver::getIntersectionPoseWithSphere src/intersection_solver.cpp:91-104 (the
 * moving-average convergence gate, :105-120, is SURVEY 8(f) "next" and not restated here).
 * Returns 1 if an intersection exists (delta > -1), pose7 = pose at t1+delta else initPose. */
int FN(orc_intersection_pose)(const TGT* tg, double t1, const double* origin, double radius,
                              double* pose7, double* delta_t) {
  double d = FN(orc_intersection_time)(tg, t1, origin, radius);
  for (int i = 0; i < 7; ++i) pose7[i] = (i == 6) ? 1.0 : 0.0;
  if (delta_t) *delta_t = d;
  if (d > -1) {
    FN(orc_target_get_pose_at)(tg, d + t1, pose7);
    return 1;
  }
  return 0;
}

/* The caller's loop over ids (test/target_manager_test.cpp:136-145, target_manager_ros.cpp:46-76)
 * with OpenMP over targets, bypassing the manager mutex.  meas is [n][7]. */
void FN(orc_batch_step)(TGT* tgs, long n, double dt, const double* meas, const unsigned char* has_meas,
                        int nthreads) {
  if (nthreads <= 0) nthreads = orc_max_threads();
#pragma omp parallel for schedule(static) num_threads(nthreads)
  for (long i = 0; i < n; ++i) {
    if (meas && (!has_meas || has_meas[i]))
      FN(orc_target_add_measurement)(&tgs[i], dt, meas + 7 * i);
    else
      FN(orc_target_update)(&tgs[i], dt);
  }
}

void FN(orc_batch_get_state)(const TGT* tgs, long n, double* x, double* P) {
  for (long i = 0; i < n; ++i) {
    const int nn = tgs[i].n;
    FN(orc_target_get_state)(&tgs[i], x ? x + i * nn : 0, P ? P + i * nn * nn : 0);
  }
}

#undef TGT
#undef FN
#undef ORC_CAT
#undef ORC_CAT_
#undef R_PI
