#!/usr/bin/env python3
"""High-precision known answers for the four motion models: steps 1-4 of one target each, evaluated in 50-digit arithmetic
(mpmath) from formulas typed from the REFERENCE sources -- not from oracle/ -- so that "the oracle, the NumPy twin and the
kernels share a misreading" and "they differ by rounding" can be told apart.

    python tests/golden/make_highprec_kat.py        ->  tests/golden/highprec_kat.npz

What is evaluated, with the reference lines it follows:
  predict   x- = A x (linear) or f(x) (EKF);  P- = A P A^T + Q                      src/kalman.cpp:84-88, :129-133
  estimate  K = P- C^T (C P- C^T + R)^-1;  x+ = x- + K (y - C x-);  P+ = (I - K C) P-   src/kalman.cpp:90-95, :135-140
  A(dt)     I + dt on the n/2 (n/3) diagonal + dt^2/2 on the 2n/3 diagonal           src/types/uniform_velocity.cpp:90-96,
                                                                                     uniform_acceleration.cpp:91-99, angular_rates.cpp:108-115
  EKF       A = [I 0 dt I 0; 0 Jrpy 0 Jomega; 0 0 I 0; 0 0 0 I] at the target's own previous posterior,
            f = [p + dt v; rpy + dt Einv(rpy) omega; v; omega], h = first six states   src/types/angular_velocities.cpp:116-150,
                                                                                     include/target_estimation/geometry.hpp:359-426
  measurement (angular models): quaternion normalised, quatToRpy, unwrap against the previous unwrapped angles
                                                                                     angular_rates.cpp:81-88, geometry.hpp:31-76, :154-176
  initial state: position and rpy of p0, zero rates; P = P0                           src/types/*.cpp constructors, geometry.hpp:619-628
The reference never initialises the unwrap memory (meas_rpy_internal_); zero is used here, as everywhere in this
repository.  Inputs are doubles (the YAML matrices as parsed to double, dt = 1/250, the poses below); every operation after that
is carried out in 50 digits and the results are rounded once, to double, at the end.
"""
import os

import numpy as np
import yaml
from mpmath import mp, mpf, matrix, sin, cos, atan2, asin, sqrt, floor, pi as mp_pi_f

mp.dps = 50
AHEAD = 0.0625              # seconds the extrapolating getters look ahead (exact in binary)
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
M_PI = mpf(float(np.pi))    # the reference's M_PI is the double nearest to pi


def c_fmod(x, y):           # C fmod: x - trunc(x / y) * y, sign of x
    q = x / y
    t = floor(q) if q >= 0 else -floor(-q)
    return x - t * y


def constrain_angle(x):     # geometry.hpp:31-36
    x = c_fmod(x + M_PI, 2 * M_PI)
    if x < 0:
        x += 2 * M_PI
    return x - M_PI


def angle_conv(a):          # geometry.hpp:43-45
    return c_fmod(constrain_angle(a), 2 * M_PI)


def angle_diff(a, b):       # geometry.hpp:53-58
    d = c_fmod(b - a + M_PI, 2 * M_PI)
    if d < 0:
        d += 2 * M_PI
    return d - M_PI


def unwrap(prev, new):      # geometry.hpp:70-76
    return prev - angle_diff(new, angle_conv(prev))


def quat_to_rpy(q):         # geometry.hpp:154-176, q = [x y z w], normalised by the caller
    x, y, z, w = q
    sp = -2 * (x * z - w * y)
    if sp > mpf("0.9999"):
        return [mpf(0), M_PI / 2, 2 * atan2(z, w)]
    if sp < mpf("-0.9999"):
        return [mpf(0), -M_PI / 2, 2 * atan2(z, w)]
    return [atan2(2 * (y * z + w * x), w * w - x * x - y * y + z * z), asin(sp), atan2(2 * (x * y + w * z), w * w + x * x - y * y - z * z)]


def normalised(q):          # Eigen::Quaterniond::normalize
    n = sqrt(sum(c * c for c in q))
    return [c / n for c in q]


def pose7_to_meas6(pose7, memory):
    """xyz + unwrapped rpy of a measured pose (angular_rates.cpp:81-88); returns (y6, new memory)"""
    rpy = quat_to_rpy(normalised([mpf(v) for v in pose7[3:7]]))
    un = [unwrap(memory[i], rpy[i]) for i in range(3)]
    return [mpf(v) for v in pose7[0:3]] + un, un


def transition_linear(n, nb, dt):
    A = mp.eye(n)
    k = n // nb
    for r in range(n - k):
        A[r, r + k] = dt
    if nb == 3:
        for r in range(n - 2 * k):
            A[r, r + 2 * k] = mpf("0.5") * dt * dt
    return A


def ekf_transition(x, dt):  # angular_velocities.cpp:116-124 with geometry.hpp:394-426
    r, p = x[3], x[4]
    wy, wz = x[10], x[11]
    cr, sr, cp, sp = cos(r), sin(r), cos(p), sin(p)
    A = mp.zeros(12)
    for i in range(3):
        A[i, i] = 1
        A[i, 6 + i] = dt
        A[6 + i, 6 + i] = 1
        A[9 + i, 9 + i] = 1
    J = [[(dt * (wy * cr * sp - wz * sp * sr)) / cp + 1, (dt * (wz * cr + wy * sr)) / (cp * cp), 0],
         [-dt * (wz * cr + wy * sr), 1, 0],
         [(dt * (wy * cr - wz * sr)) / cp, (dt * sp * (wz * cr + wy * sr)) / (cp * cp), 1]]
    W = [[dt, (dt * sp * sr) / cp, (dt * cr * sp) / cp],
         [0, dt * cr, -dt * sr],
         [0, (dt * sr) / cp, (dt * cr) / cp]]
    for i in range(3):
        for j in range(3):
            A[3 + i, 3 + j] = J[i][j]
            A[3 + i, 9 + j] = W[i][j]
    return A


def ekf_f(x, dt):           # angular_velocities.cpp:126-140 with geometry.hpp:359-374
    r, p = x[3], x[4]
    cr, sr, cp, sp = cos(r), sin(r), cos(p), sin(p)
    E = [[1, (sp * sr) / cp, (cr * sp) / cp], [0, cr, -sr], [0, sr / cp, cr / cp]]
    out = [mpf(0)] * 12
    for i in range(3):
        out[i] = x[i] + dt * x[6 + i]
        out[3 + i] = x[3 + i] + dt * sum(E[i][j] * x[9 + j] for j in range(3))
        out[6 + i] = x[6 + i]
        out[9 + i] = x[9 + i]
    return out


# ---- derived outputs: what updateTargetState leaves in T_, twist_, acceleration_, pose_internal_, and the
# getters that extrapolate to t1 = t + ahead
def rpy_to_quat(rpy):       # geometry.hpp:178-189 (x y z w), normalised
    ph, th, ps = rpy[0] / 2, rpy[1] / 2, rpy[2] / 2
    w = cos(ph) * cos(th) * cos(ps) + sin(ph) * sin(th) * sin(ps)
    x = sin(ph) * cos(th) * cos(ps) - cos(ph) * sin(th) * sin(ps)
    y = cos(ph) * sin(th) * cos(ps) + sin(ph) * cos(th) * sin(ps)
    z = cos(ph) * cos(th) * sin(ps) - sin(ph) * sin(th) * cos(ps)
    return normalised([x, y, z, w])


def quat_to_rot(q):         # Eigen::Quaterniond::toRotationMatrix
    x, y, z, w = q
    return [[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
            [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
            [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]]


def rot_to_quat(m):         # Eigen's Matrix3 -> Quaternion (isometryToPose7d, geometry.hpp:590-594): trace branch, else largest diagonal
    t = m[0][0] + m[1][1] + m[2][2]
    if t > 0:
        t = sqrt(t + 1)
        w = t / 2
        t = mpf("0.5") / t
        return [(m[2][1] - m[1][2]) * t, (m[0][2] - m[2][0]) * t, (m[1][0] - m[0][1]) * t, w]
    i = 0
    if m[1][1] > m[0][0]:
        i = 1
    if m[2][2] > m[i][i]:
        i = 2
    j, k = (i + 1) % 3, (i + 2) % 3
    t = sqrt(m[i][i] - m[j][j] - m[k][k] + 1)
    q = [mpf(0)] * 4
    q[i] = t / 2
    t = mpf("0.5") / t
    q[3] = (m[k][j] - m[j][k]) * t
    q[j] = (m[j][i] + m[i][j]) * t
    q[k] = (m[k][i] + m[i][k]) * t
    return q


def rot_to_rpy(m):          # geometry.hpp:191-196
    return [atan2(m[2][1], m[2][2]), atan2(-m[2][0], sqrt(m[2][1] * m[2][1] + m[2][2] * m[2][2])), atan2(m[1][0], m[0][0])]


def outputs(model, x, ahead):
    """(pose7, twist6, acc6) at the target's own time, and (pose7, twist6) extrapolated by `ahead` seconds -- updateTargetState and
    getEstimatedPose / getEstimatedTwist(t1) of src/types/*.cpp."""
    z3 = [mpf(0)] * 3
    pos = list(x[0:3])
    if model == "uniform_velocity":                      # uniform_velocity.cpp:98-140
        tw, ac = list(x[3:6]) + z3, z3 + z3
        return (pos + [0, 0, 0, 1], tw, ac), ([pos[i] + tw[i] * ahead for i in range(3)] + [0, 0, 0, 1], tw)
    if model == "uniform_acceleration":                  # uniform_acceleration.cpp:100-140
        tw, ac = list(x[3:6]) + z3, list(x[6:9]) + z3
        return (pos + [0, 0, 0, 1], tw, ac), ([pos[i] + tw[i] * ahead + mpf("0.5") * ac[i] * ahead * ahead for i in range(3)] + [0, 0, 0, 1],
                                            [tw[i] + ac[i] * ahead for i in range(6)])
    Rm = quat_to_rot(rpy_to_quat(list(x[3:6])))          # T_.linear()
    pose7 = pos + rot_to_quat(Rm)
    rpy_i = rot_to_rpy(Rm)                               # pose_internal_ (isometryToPose6d)
    if model == "angular_rates":                         # angular_rates.cpp:117-160
        cr, sr, cp, sp = cos(rpy_i[0]), sin(rpy_i[0]), cos(rpy_i[1]), sin(rpy_i[1])
        Ear = [[1, 0, -sp], [0, cr, cp * sr], [0, -sr, cp * cr]]     # geometry.hpp:333-352
        rates = list(x[9:12])
        tw = list(x[6:9]) + [sum(Ear[i][j] * rates[j] for j in range(3)) for i in range(3)]
        ac = list(x[12:18])
        p6 = pos + rpy_i
        v6 = [p6[i] + tw[i] * ahead + mpf("0.5") * ac[i] * ahead * ahead for i in range(6)]
        return (pose7, tw, ac), (v6[0:3] + normalised(rpy_to_quat(v6[3:6])), [tw[i] + ac[i] * ahead for i in range(6)])
    # angular_velocities.cpp:152-184: position advanced by v, the quaternion by Qtran(ahead, omega) (geometry.hpp:493-504)
    tw, ac = list(x[6:12]), z3 + z3
    om = tw[3:6]
    q = rpy_to_quat(rpy_i)
    on = sqrt(sum(c * c for c in om))
    if on > 0:
        half = on * ahead / 2
        S = [[0, -om[2], om[1], om[0]], [om[2], 0, -om[0], om[1]], [-om[1], om[0], 0, om[2]], [-om[0], -om[1], -om[2], 0]]   # omegaToMatrix: 0.5 * this
        Qt = [[(cos(half) if i == j else 0) + 2 / on * sin(half) * mpf("0.5") * S[i][j] for j in range(4)] for i in range(4)]
        q = [sum(Qt[i][j] * q[j] for j in range(4)) for i in range(4)]
    return (pose7, tw, ac), ([pos[i] + tw[i] * ahead for i in range(3)] + normalised(q), tw)


def run(model, Q, R, P0, p0, dt, stream):
    n, m = Q.rows, R.rows
    angular = model in ("angular_rates", "angular_velocities")
    x = [mpf(0)] * n
    for i in range(3):
        x[i] = mpf(p0[i])
    if angular:
        rpy0 = quat_to_rpy(normalised([mpf(v) for v in p0[3:7]]))   # pose7dToPose6d, geometry.hpp:619-628
        for i in range(3):
            x[3 + i] = rpy0[i]
    P = P0.copy()
    memory = [mpf(0)] * 3
    C = mp.zeros(m, n)
    for i in range(m):
        C[i, i] = 1
    I = mp.eye(n)
    xs, Ps, outs = [], [], []
    for meas in stream:
        if model == "angular_velocities":
            A = ekf_transition(x, dt)
            xp = ekf_f(x, dt)
        else:
            A = transition_linear(n, 2 if model == "uniform_velocity" else 3, dt)
            xp = list(A * matrix(x))
        P = A * P * A.T + Q
        if meas is not None:
            if angular:
                y, memory = pose7_to_meas6(meas, memory)
            else:
                y = [mpf(v) for v in meas[0:3]]
            S = C * P * C.T + R
            K = P * C.T * (S ** -1)
            innov = matrix([y[i] - xp[i] for i in range(m)])
            xp = list(matrix(xp) + K * innov)
            P = (I - K * C) * P
        x = xp
        xs.append([float(v) for v in x])
        Ps.append([[float(P[i, j]) for j in range(n)] for i in range(n)])
        now, later = outputs(model, x, mpf(AHEAD))
        outs.append([float(v) for v in now[0] + now[1] + now[2] + later[0] + later[1]])    # pose7 twist6 acc6 | pose7 twist6 at t + AHEAD
    return np.array(xs), np.array(Ps), np.array(outs)


def rpy_quat(r, p, y):      # only to WRITE test inputs (plain double arithmetic; the result is the input)
    cr, sr, cp, sp, cy, sy = np.cos(r / 2), np.sin(r / 2), np.cos(p / 2), np.sin(p / 2), np.cos(y / 2), np.sin(y / 2)
    return [sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy]


def intersection_cases():
    """Sphere intersection of uniform-acceleration targets straight after their creation (state = [p0 v0 a0] exactly, t1 = t0):
    the quartic of src/intersection_solver.cpp:58-76, its roots to 50 digits (mpmath.polyroots), the reference's selection rule
    (Solver::lowestRealRoot, :4-17 with Eigen's smallestRealRoot: the smallest real part among the roots with |imag| < 1e-10; none,
    or a negative one, or a zero leading coefficient -> -1) and the pose at t1 + delta (uniform_acceleration.cpp:120-131).
    Cases are kept away from tangency: there a companion-matrix solver in double classifies by its own rounding."""
    from mpmath import polyroots
    rng = np.random.default_rng(20240808)
    origin, radius = np.array([0.25, -0.5, 0.125]), 1.5
    P, V, A = [], [], []
    for k in range(48):
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        dist = rng.uniform(2.0, 9.0) if k % 4 else rng.uniform(0.2, 1.2)      # every fourth target starts INSIDE the sphere
        p = origin + d * dist
        v = -d * rng.uniform(0.5, 6.0) + rng.normal(0, 0.15, 3)
        a = rng.normal(0, 0.4, 3) + np.array([0, 0, -0.5])
        if k % 12 == 5:
            a = np.zeros(3)                                                    # zero acceleration: "no intersection" in the reference
        if k % 12 == 7:
            v = d * rng.uniform(1.0, 4.0); a = d * rng.uniform(0.5, 2.0)       # flying away: no real root >= 0
        P.append(p); V.append(v); A.append(a)
    P, V, A = np.array(P), np.array(V), np.array(A)
    delta, pose, margin = [], [], []
    for p, v, a in zip(P, V, A):
        x, y, z = [mpf(float(c)) - mpf(float(o)) for c, o in zip(p, origin)]
        vx, vy, vz = [mpf(float(c)) for c in v]
        ax, ay, az = [mpf(float(c)) for c in a]
        R = mpf(radius)
        c4 = mpf("0.25") * (ax * ax + ay * ay + az * az)
        c3 = vx * ax + vy * ay + vz * az
        c2 = vx * vx + vy * vy + vz * vz + x * ax + y * ay + z * az
        c1 = 2 * (x * vx + y * vy + z * vz)
        c0 = x * x + y * y + z * z - R * R
        if c4 == 0:
            delta.append(-1.0); pose.append([0, 0, 0, 0, 0, 0, 1.0]); margin.append(1.0)
            continue
        roots = polyroots([c4, c3, c2, c1, c0], maxsteps=200, extraprec=200)
        real = [r.real for r in roots if abs(r.imag) < mpf("1e-10")]
        # distance from a classification boundary: |imag| of the complex roots, |value| of the smallest real one, gap between real roots
        m = min([abs(r.imag) for r in roots if abs(r.imag) >= mpf("1e-10")] + [mpf(1)])
        if real:
            rs = sorted(real)
            m = min([m, abs(rs[0])] + [rs[i + 1] - rs[i] for i in range(len(rs) - 1)])
        margin.append(float(m))
        dmin = min(real) if real else mpf(-1)
        if dmin < 0:
            delta.append(-1.0); pose.append([0, 0, 0, 0, 0, 0, 1.0])
        else:
            delta.append(float(dmin))
            pose.append([float(mpf(float(p[i])) + mpf(float(v[i])) * dmin + mpf("0.5") * mpf(float(a[i])) * dmin * dmin) for i in range(3)] + [0, 0, 0, 1.0])
    return dict(ix_origin=origin, ix_radius=np.array(radius), ix_p0=P, ix_v0=V, ix_a0=A, ix_delta=np.array(delta), ix_pose=np.array(pose),
                ix_margin=np.array(margin))


def main():
    dt = 1.0 / 250.0
    # one target per model: initial pose, then four ticks: measured, measured, predict only, measured.  The yaw of the
    # measurements crosses +pi between ticks 1 and 2 (the unwrap must carry it), the quaternions are not unit length
    p0 = np.array([0.3, -0.2, 0.1] + [2.0 * c for c in rpy_quat(0.1, -0.2, 3.0)])
    stream = [np.array([0.3105, -0.1958, 0.1012] + [1.5 * c for c in rpy_quat(0.11, -0.19, 3.1)]),
              np.array([0.3192, -0.1911, 0.1043] + [0.7 * c for c in rpy_quat(0.12, -0.18, -3.08)]),
              None,
              np.array([0.3391, -0.1832, 0.1077] + [1.1 * c for c in rpy_quat(0.15, -0.16, -2.9)])]
    out = {"dt": np.array(dt), "ahead": np.array(AHEAD), "p0": p0, "has": np.array([s is not None for s in stream]),
           "meas": np.array([s if s is not None else np.zeros(7) for s in stream])}
    for model in ("uniform_velocity", "uniform_acceleration", "angular_rates", "angular_velocities"):
        y = yaml.safe_load(open(os.path.join(ROOT, "models", "model_%s_params.yaml" % model)))
        n = {"uniform_velocity": 6, "uniform_acceleration": 9, "angular_rates": 18, "angular_velocities": 12}[model]
        m = 3 if n < 12 else 6
        Q = matrix(n, n); R = matrix(m, m); P0 = matrix(n, n)
        for i in range(n):
            for j in range(n):
                Q[i, j] = mpf(float(y["Q"][i * n + j]))
                P0[i, j] = mpf(float(y["P"][i * n + j]))
        for i in range(m):
            for j in range(m):
                R[i, j] = mpf(float(y["R"][i * m + j]))
        xs, Ps, outs = run(model, Q, R, P0, p0, mpf(dt), stream)
        out["x_" + model] = xs
        out["P_" + model] = Ps
        out["out_" + model] = outs      # [tick][pose7 twist6 acc6 | pose7 twist6 extrapolated by `ahead`]
        print(model, "x after tick 4:", xs[-1][:6])
    # the gimbal branches of quatToRpy (geometry.hpp:156-169, |sin pitch| > 0.9999) through the angular-rates model (linear in rpy;
    # the EKF's Jacobians are singular there and are not driven through it): pitch +pi/2, then -pi/2, then back to a regular attitude
    gstream = [np.array([0.31, -0.19, 0.10] + [1.3 * c for c in rpy_quat(0.4, np.pi / 2 - 1e-3, -0.7)]),
               np.array([0.32, -0.18, 0.11] + [0.9 * c for c in rpy_quat(-0.3, -np.pi / 2 + 2e-3, 0.5)]),
               np.array([0.33, -0.17, 0.12] + [1.0 * c for c in rpy_quat(0.2, 0.3, 0.9)])]
    for g in gstream[:2]:
        q = np.array(g[3:7]) / np.linalg.norm(g[3:7])
        assert abs(-2 * (q[0] * q[2] - q[3] * q[1])) > 0.9999          # really inside the branch
    y = yaml.safe_load(open(os.path.join(ROOT, "models", "model_angular_rates_params.yaml")))
    Q = matrix(18, 18); R = matrix(6, 6); P0 = matrix(18, 18)
    for i in range(18):
        for j in range(18):
            Q[i, j] = mpf(float(y["Q"][i * 18 + j])); P0[i, j] = mpf(float(y["P"][i * 18 + j]))
    for i in range(6):
        for j in range(6):
            R[i, j] = mpf(float(y["R"][i * 6 + j]))
    xs, Ps, outs = run("angular_rates", Q, R, P0, p0, mpf(dt), gstream)
    out.update(gimbal_meas=np.array(gstream), gimbal_x=xs, gimbal_P=Ps, gimbal_out=outs)
    print("gimbal branches: rpy after the three ticks", xs[:, 3:6])
    ix = intersection_cases()
    out.update(ix)
    print("intersection cases:", len(ix["ix_delta"]), "hits:", int((ix["ix_delta"] > -1).sum()), "smallest margin to a classification boundary: %.3g" % ix["ix_margin"].min())
    np.savez(os.path.join(HERE, "highprec_kat.npz"), **out)


if __name__ == "__main__":
    main()
