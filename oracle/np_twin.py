"""NumPy twin of the reference's per-target Kalman path -- TEST INFRASTRUCTURE ONLY.

An independent second restatement (written from the reference sources, not from the C
oracle) used to cross-check oracle/te_oracle.c and to generate the committed fixtures under
tests/golden/ (tests/golden/make_golden.py).  float64 only, BLAS matmul (so summation order
and FMA use differ from the C oracle: agreement is to rounding, not bitwise).

Citations are file:line under /root/reference.
"""
import math

import numpy as np

ANGULAR_RATES, ANGULAR_VELOCITIES, UNIFORM_ACCELERATION, UNIFORM_VELOCITY = 0, 1, 2, 3
DIMS = {ANGULAR_RATES: (18, 6), ANGULAR_VELOCITIES: (12, 6), UNIFORM_ACCELERATION: (9, 3),
        UNIFORM_VELOCITY: (6, 3)}
TWO_PI = 2 * math.pi


# ---- include/target_estimation/geometry.hpp -------------------------------------------------
def constrain_angle(x):  # :31-36
    x = math.fmod(x + math.pi, TWO_PI)
    if x < 0:
        x += TWO_PI
    return x - math.pi


def angle_conv(a):  # :43-45
    return math.fmod(constrain_angle(a), TWO_PI)


def angle_diff(a, b):  # :53-58
    d = math.fmod(b - a + math.pi, TWO_PI)
    if d < 0:
        d += TWO_PI
    return d - math.pi


def unwrap(prev, new):  # :70-76
    return np.array([prev[i] - angle_diff(new[i], angle_conv(prev[i])) for i in range(3)])


def quat_normalize(q):
    return q / math.sqrt(float(q @ q))


def quat_to_rpy(q):  # :154-176, q = [x y z w]
    x, y, z, w = q
    s = -2 * (x * z - w * y)
    if s > 0.9999:
        return np.array([0.0, math.pi / 2, 2 * math.atan2(z, w)])
    if s < -0.9999:
        return np.array([0.0, -math.pi / 2, 2 * math.atan2(z, w)])
    return np.array([math.atan2(2 * (y * z + w * x), w * w - x * x - y * y + z * z),
                     math.asin(s),
                     math.atan2(2 * (x * y + w * z), w * w + x * x - y * y - z * z)])


def rpy_to_quat(rpy):  # :178-189
    phi, the, psi = rpy[0] / 2, rpy[1] / 2, rpy[2] / 2
    c, s = math.cos, math.sin
    w = c(phi) * c(the) * c(psi) + s(phi) * s(the) * s(psi)
    x = s(phi) * c(the) * c(psi) - c(phi) * s(the) * s(psi)
    y = c(phi) * s(the) * c(psi) + s(phi) * c(the) * s(psi)
    z = c(phi) * c(the) * s(psi) - s(phi) * s(the) * c(psi)
    return quat_normalize(np.array([x, y, z, w]))


def rot_to_rpy(R):  # :191-196
    return np.array([math.atan2(R[2, 1], R[2, 2]),
                     math.atan2(-R[2, 0], math.sqrt(R[2, 1] ** 2 + R[2, 2] ** 2)),
                     math.atan2(R[1, 0], R[0, 0])])


def quat_to_rot(q):  # Eigen::Quaterniond::toRotationMatrix
    x, y, z, w = q
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return np.array([[1 - (tyy + tzz), txy - twz, txz + twy],
                     [txy + twz, 1 - (txx + tzz), tyz - twx],
                     [txz - twy, tyz + twx, 1 - (txx + tyy)]])


def rot_to_quat(R):  # Eigen matrix -> quaternion (used by isometryToPose7d :590-594)
    q = np.zeros(4)
    t = R[0, 0] + R[1, 1] + R[2, 2]
    if t > 0:
        t = math.sqrt(t + 1.0)
        q[3] = 0.5 * t
        t = 0.5 / t
        q[0] = (R[2, 1] - R[1, 2]) * t
        q[1] = (R[0, 2] - R[2, 0]) * t
        q[2] = (R[1, 0] - R[0, 1]) * t
    else:
        i = 0
        if R[1, 1] > R[0, 0]:
            i = 1
        if R[2, 2] > R[i, i]:
            i = 2
        j = (i + 1) % 3
        k = (j + 1) % 3
        t = math.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        q[i] = 0.5 * t
        t = 0.5 / t
        q[3] = (R[k, j] - R[j, k]) * t
        q[j] = (R[j, i] + R[i, j]) * t
        q[k] = (R[k, i] + R[i, k]) * t
    return q


def ear_base(rpy):  # :333-351
    cr, sr, cp, sp = math.cos(rpy[0]), math.sin(rpy[0]), math.cos(rpy[1]), math.sin(rpy[1])
    return np.array([[1, 0, -sp], [0, cr, cp * sr], [0, -sr, cp * cr]])


def ear_base_inv(rpy):  # :359-374
    cr, sr, cp, sp = math.cos(rpy[0]), math.sin(rpy[0]), math.cos(rpy[1]), math.sin(rpy[1])
    return np.array([[1, (sp * sr) / cp, (cr * sp) / cp], [0, cr, -sr], [0, sr / cp, cr / cp]])


def jac_rpy(rpy, omega, dt):  # :394-410
    wy, wz = omega[1], omega[2]
    cr, sr, cp, sp = math.cos(rpy[0]), math.sin(rpy[0]), math.cos(rpy[1]), math.sin(rpy[1])
    return np.array([
        [(dt * (wy * cr * sp - wz * sp * sr)) / cp + 1, (dt * (wz * cr + wy * sr)) / (cp * cp), 0],
        [-dt * (wz * cr + wy * sr), 1, 0],
        [(dt * (wy * cr - wz * sr)) / cp, (dt * sp * (wz * cr + wy * sr)) / (cp * cp), 1]])


def jac_omega(rpy, dt):  # :412-426
    cr, sr, cp, sp = math.cos(rpy[0]), math.sin(rpy[0]), math.cos(rpy[1]), math.sin(rpy[1])
    return np.array([[dt, (dt * sp * sr) / cp, (dt * cr * sp) / cp],
                     [0, dt * cr, -dt * sr],
                     [0, (dt * sr) / cp, (dt * cr) / cp]])


def qtran(dt, omega):  # :448-465, :493-504
    nrm = math.sqrt(float(omega @ omega))
    S = 0.5 * np.array([[0, -omega[2], omega[1], omega[0]],
                        [omega[2], 0, -omega[0], omega[1]],
                        [-omega[1], omega[0], 0, omega[2]],
                        [-omega[0], -omega[1], -omega[2], 0]])
    if nrm > 0:
        tmp = nrm * dt / 2.0
        return math.cos(tmp) * np.eye(4) + 2.0 / nrm * math.sin(tmp) * S
    return np.eye(4)


def pose7_to_pose6(p7):  # :619-628
    return np.concatenate([p7[:3], quat_to_rpy(quat_normalize(p7[3:7]))])


# ---- target (src/types/*.cpp + src/kalman.cpp) ------------------------------------------------
class Target:
    def __init__(self, model, Q, R, P0, p0, dt0, t0=0.0, v0=None, a0=None):
        self.model = model
        self.n, self.m = DIMS[model]
        n, m = self.n, self.m
        self.Q, self.R = np.array(Q, float), np.array(R, float)
        self.P = np.array(P0, float).copy()
        self.t, self.n_meas = float(t0), 0
        v0 = np.zeros(6) if v0 is None else np.asarray(v0, float)
        a0 = np.zeros(6) if a0 is None else np.asarray(a0, float)
        p0 = np.asarray(p0, float)
        self.C = np.zeros((m, n))
        self.C[np.arange(m), np.arange(m)] = 1.0
        self.meas_rpy = np.zeros(3)  # zero-initialised by choice (never set in the reference)
        x = np.zeros(n)
        if model == UNIFORM_VELOCITY:  # uniform_velocity.cpp:51-54
            x[0:3], x[3:6] = p0[:3], v0[:3]
        elif model == UNIFORM_ACCELERATION:  # uniform_acceleration.cpp:51-55
            x[0:3], x[3:6], x[6:9] = p0[:3], v0[:3], a0[:3]
        elif model == ANGULAR_RATES:  # angular_rates.cpp:58-63
            x[0:6], x[6:12], x[12:18] = pose7_to_pose6(p0), v0, a0
        else:  # angular_velocities.cpp:52-56
            x[0:6], x[6:12] = pose7_to_pose6(p0), v0
        self.x = x
        self.acceleration = np.zeros(6)
        self._outputs()

    def _A(self, dt):
        n = self.n
        A = np.eye(n)
        if self.model == UNIFORM_VELOCITY:  # uniform_velocity.cpp:90-96
            A[np.arange(3), np.arange(3) + 3] = dt
        elif self.model in (UNIFORM_ACCELERATION, ANGULAR_RATES):  # :91-99 / angular_rates.cpp:108-115
            k = n // 3
            A[np.arange(2 * k), np.arange(2 * k) + k] = dt
            A[np.arange(k), np.arange(k) + 2 * k] = 0.5 * dt * dt
        else:  # angular_velocities.cpp:116-124
            A[0:3, 6:9] = np.eye(3) * dt
            A[3:6, 3:6] = jac_rpy(self.x[3:6], self.x[9:12], dt)
            A[3:6, 9:12] = jac_omega(self.x[3:6], dt)
        return A

    def _f(self, x, dt):  # angular_velocities.cpp:126-140
        out = np.zeros(12)
        out[0:3] = x[0:3] + dt * x[6:9]
        out[6:9], out[9:12] = x[6:9], x[9:12]
        out[3:6] = x[3:6] + (dt * ear_base_inv(x[3:6])) @ x[9:12]
        return out

    def _step(self, dt, y):
        A = self._A(dt)
        # predict, src/kalman.cpp:84-88 / :129-133
        xn = self._f(self.x, dt) if self.model == ANGULAR_VELOCITIES else A @ self.x
        P = A @ self.P @ A.T + self.Q
        if y is not None:  # estimate, src/kalman.cpp:90-95 / :135-140
            C = self.C
            K = P @ C.T @ np.linalg.inv(C @ P @ C.T + self.R)
            xn = xn + K @ (y - C @ xn)
            P = (np.eye(self.n) - K @ C) @ P
        self.x, self.P = xn, P
        self._outputs()
        self.t += dt

    def _outputs(self):  # updateTargetState of each model
        x = self.x
        self.trans = x[0:3].copy()
        self.twist = np.zeros(6)
        if self.model in (UNIFORM_VELOCITY, UNIFORM_ACCELERATION):
            self.Rm = np.eye(3)
            self.twist[0:3] = x[3:6]
            self.acceleration = np.zeros(6)
            if self.model == UNIFORM_ACCELERATION:
                self.acceleration[0:3] = x[6:9]
        elif self.model == ANGULAR_RATES:  # angular_rates.cpp:117-138
            self.Rm = quat_to_rot(rpy_to_quat(x[3:6]))
            self.twist[0:3] = x[6:9]
            self.twist[3:6] = ear_base(rot_to_rpy(self.Rm)) @ x[9:12]
            self.acceleration = x[12:18].copy()
        else:  # angular_velocities.cpp:153-169
            self.Rm = quat_to_rot(rpy_to_quat(x[3:6]))
            self.twist = x[6:12].copy()
        self.pose_internal = np.concatenate([self.trans, rot_to_rpy(self.Rm)])

    def add_measurement(self, dt, meas):
        meas = np.asarray(meas, float)
        self.n_meas += 1
        if self.model in (UNIFORM_VELOCITY, UNIFORM_ACCELERATION):
            y = meas[0:3].copy()
        else:  # angular_rates.cpp:81-88 / angular_velocities.cpp:89-96
            rpy = unwrap(self.meas_rpy, quat_to_rpy(quat_normalize(meas[3:7])))
            self.meas_rpy = rpy
            y = np.concatenate([meas[0:3], rpy])
        self._step(dt, y)

    def update(self, dt):
        self._step(dt, None)

    def pose(self):  # target_interface.cpp:100-104
        return np.concatenate([self.trans, rot_to_quat(self.Rm)])

    def pose_at(self, t1):
        d = t1 - self.t
        if self.model == UNIFORM_VELOCITY:
            return np.concatenate([self.trans + self.twist[:3] * d, [0, 0, 0, 1.0]])
        if self.model == UNIFORM_ACCELERATION:
            return np.concatenate([self.trans + self.twist[:3] * d + 0.5 * self.acceleration[:3] * d * d,
                                   [0, 0, 0, 1.0]])
        if self.model == ANGULAR_RATES:
            v6 = self.pose_internal + self.twist * d + 0.5 * self.acceleration * d * d
            return np.concatenate([v6[:3], quat_normalize(rpy_to_quat(v6[3:6]))])
        q = quat_normalize(qtran(d, self.twist[3:6]) @ rpy_to_quat(self.pose_internal[3:6]))
        return np.concatenate([self.trans + self.twist[:3] * d, q])

    def twist_at(self, t1):
        if self.model in (UNIFORM_ACCELERATION, ANGULAR_RATES):
            return self.twist + self.acceleration * (t1 - self.t)
        return self.twist.copy()

    def intersection_time(self, t1, origin, radius):  # src/intersection_solver.cpp:42-89
        r = self.pose_at(t1)[:3] - np.asarray(origin, float)
        v, a = self.twist_at(t1)[:3], self.acceleration[:3]
        c = [r @ r - radius * radius, 2 * (r @ v), v @ v + r @ a, v @ a, 0.25 * (a @ a)]
        return lowest_real_root(c)


def lowest_real_root(c):  # src/intersection_solver.cpp:4-17 (companion-matrix eigenvalues)
    if not abs(c[-1]) > 0.0:
        return -1.0
    roots = np.roots(np.asarray(c, float)[::-1])
    real = [z.real for z in roots if abs(z.imag) < 1e-10]
    if not real:
        return -1.0
    r = min(real)
    return -1.0 if r < 0 else r
