"""ctypes front-end of the C oracle (oracle/te_oracle.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")

# include/target_estimation/target_manager.hpp:38 (reference enum order)
MODELS = {"angular_rates": 0, "angular_velocities": 1, "uniform_acceleration": 2, "uniform_velocity": 3}
MODEL_DIMS = {0: (18, 6), 1: (12, 6), 2: (9, 3), 3: (6, 3)}

_libs = {}


def build(force=False):
    """Compile the oracle with the committed Makefile (gcc/g++ only)."""
    strict = os.path.join(_BUILD, "libte_oracle.so")
    fast = os.path.join(_BUILD, "libte_oracle_fast.so")
    if force or not (os.path.exists(strict) and os.path.exists(fast)):
        subprocess.check_call(["make", "-C", _HERE] + (["-B"] if force else []),
                              stdout=subprocess.DEVNULL)
    return strict, fast


def load(fast=False):
    """Return the ctypes handle (strict build = the checker; fast build = CPU baseline)."""
    key = "fast" if fast else "strict"
    if key in _libs:
        return _libs[key]
    strict_p, fast_p = build()
    lib = C.CDLL(fast_p if fast else strict_p)
    dp = C.POINTER(C.c_double)
    for sfx, real in (("f64", C.c_double), ("f32", C.c_float)):
        rp = C.POINTER(real)
        g = lambda name: getattr(lib, "%s_%s" % (name, sfx))  # noqa: E731
        g("orc_target_sizeof").restype = C.c_int
        g("orc_target_init").restype = C.c_int
        g("orc_target_init").argtypes = [C.c_void_p, C.c_int, C.c_uint, C.c_double, C.c_double,
                                         dp, dp, dp, dp, dp, dp]
        g("orc_target_add_measurement").argtypes = [C.c_void_p, C.c_double, dp]
        g("orc_target_update").argtypes = [C.c_void_p, C.c_double]
        g("orc_target_get_state").argtypes = [C.c_void_p, dp, dp]
        for nm in ("orc_target_get_pose", "orc_target_get_twist", "orc_target_get_acceleration", "orc_target_get_measured_pose",
                   "orc_target_get_pose6", "orc_target_get_transform"):
            g(nm).argtypes = [C.c_void_p, dp]
        g("orc_target_get_period_estimate").restype = C.c_double
        g("orc_target_get_period_estimate").argtypes = [C.c_void_p]
        for nm in ("orc_target_get_pose_at", "orc_target_get_twist_at",
                   "orc_target_get_acceleration_at"):
            g(nm).argtypes = [C.c_void_p, C.c_double, dp]
        g("orc_intersection_time").restype = C.c_double
        g("orc_intersection_time").argtypes = [C.c_void_p, C.c_double, dp, C.c_double]
        g("orc_intersection_pose").restype = C.c_int
        g("orc_intersection_pose").argtypes = [C.c_void_p, C.c_double, dp, C.c_double, dp, dp]
        g("orc_batch_step").argtypes = [C.c_void_p, C.c_long, C.c_double, dp,
                                        C.POINTER(C.c_ubyte), C.c_int]
        g("orc_batch_get_state").argtypes = [C.c_void_p, C.c_long, dp, dp]
        for nm in ("orc_constrain_angle", "orc_angle_conv"):
            g(nm).restype = real
            g(nm).argtypes = [real]
        for nm in ("orc_angle_diff", "orc_unwrap"):
            g(nm).restype = real
            g(nm).argtypes = [real, real]
        g("orc_quat_normalize").argtypes = [rp]
        for nm in ("orc_quat_to_rpy", "orc_rpy_to_quat", "orc_rot_to_rpy", "orc_quat_to_rot",
                   "orc_rot_to_quat", "orc_rpy_to_ear_base", "orc_rpy_to_ear_base_inv"):
            g(nm).argtypes = [rp, rp]
        g("orc_ear_base_inv_jac_rpy").argtypes = [rp, rp, real, rp]
        g("orc_ear_base_inv_jac_omega").argtypes = [rp, real, rp]
        g("orc_qtran").argtypes = [real, rp, rp]
        g("orc_inverse").restype = C.c_int
        g("orc_inverse").argtypes = [C.c_int, rp, rp]
    lib.orc_gate_sizeof.restype = C.c_int
    lib.orc_gate_init.argtypes = [C.c_void_p, C.c_int]
    lib.orc_gate_update.restype = C.c_int
    lib.orc_gate_update.argtypes = [C.c_void_p, C.c_int, dp, C.c_double, C.c_double, dp, dp]
    lib.orc_moving_avg_init.argtypes = [C.c_void_p, C.c_int]
    lib.orc_moving_avg_update.restype = C.c_double
    lib.orc_moving_avg_update.argtypes = [C.c_void_p, C.c_double]
    lib.orc_lowest_real_root.restype = C.c_double
    lib.orc_lowest_real_root.argtypes = [dp, C.c_int]
    lib.orc_poly_roots.restype = C.c_int
    lib.orc_poly_roots.argtypes = [dp, C.c_int, dp]
    lib.orc_max_threads.restype = C.c_int
    lib.orc_ref_test_stream.argtypes = [dp, C.c_int, C.c_int, C.c_double, dp, dp]
    lib.orc_harness_run_f64.argtypes = [C.c_int, dp, dp, dp, dp, C.c_long, C.c_double, dp, dp]
    lib.orc_stream_fill.argtypes = [C.c_int, C.c_ulonglong, C.c_long, C.c_long, C.c_long, C.c_long, C.c_double, C.c_double,
                                    C.c_double, C.c_int, dp, C.POINTER(C.c_ubyte), dp, dp]
    _libs[key] = lib
    return lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _d(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        assert a.shape == tuple(shape), (a.shape, shape)
    return a


def load_model_yaml(path):
    """Dependency-light reader of the reference's model files (models/*.yaml):
    keys type / frequency / Q / R / P, matrices as flat row-major flow sequences
    (src/target_manager.cpp:18-104)."""
    import yaml
    with open(path) as f:
        node = yaml.safe_load(f)
    model = MODELS[node["type"]]
    n, m = MODEL_DIMS[model]
    return dict(model=model, type=node["type"], frequency=float(node["frequency"]),
                Q=np.array(node["Q"], dtype=np.float64).reshape(n, n),
                R=np.array(node["R"], dtype=np.float64).reshape(m, m),
                P=np.array(node["P"], dtype=np.float64).reshape(n, n))


class OracleBatch:
    """An array of N independent oracle targets of one model (the caller's loop over ids)."""

    def __init__(self, model, Q, R, P0, p0, dt0, t0=0.0, v0=None, a0=None, dtype="f64", fast=False):
        self.lib = load(fast)
        self.sfx = dtype
        self.model = int(model)
        self.n, self.m = MODEL_DIMS[self.model]
        p0 = _d(p0).reshape(-1, 7)
        self.N = p0.shape[0]
        self.size = self._f("orc_target_sizeof")()
        self.buf = (C.c_char * (self.size * self.N))()
        self.base = C.addressof(self.buf)
        Q = _d(Q, (self.n, self.n)); R = _d(R, (self.m, self.m))
        P0 = _d(P0)
        per_target_P0 = P0.ndim == 3
        v0 = None if v0 is None else _d(v0).reshape(self.N, 6)
        a0 = None if a0 is None else _d(a0).reshape(self.N, 6)
        init = self._f("orc_target_init")
        null = C.POINTER(C.c_double)()
        for i in range(self.N):
            P0i = P0[i] if per_target_P0 else P0
            rc = init(self._at(i), self.model, i, float(dt0), float(t0), _dp(Q), _dp(R),
                      _dp(np.ascontiguousarray(P0i)), _dp(p0[i]),
                      _dp(v0[i]) if v0 is not None else null,
                      _dp(a0[i]) if a0 is not None else null)
            assert rc == 0

    def _f(self, name):
        return getattr(self.lib, "%s_%s" % (name, self.sfx))

    def _at(self, i):
        return C.c_void_p(self.base + i * self.size)

    def step(self, dt, meas=None, has_meas=None, nthreads=1):
        """meas [N,7] doubles or None (predict-only); has_meas [N] uint8 or None."""
        mp = C.POINTER(C.c_double)()
        if meas is not None:
            meas = _d(meas, (self.N, 7))
            mp = _dp(meas)
        hp = C.POINTER(C.c_ubyte)()
        if has_meas is not None:
            has_meas = np.ascontiguousarray(has_meas, dtype=np.uint8)
            hp = has_meas.ctypes.data_as(C.POINTER(C.c_ubyte))
        self._f("orc_batch_step")(self._at(0), self.N, float(dt), mp, hp, int(nthreads))

    def state(self):
        x = np.empty((self.N, self.n)); P = np.empty((self.N, self.n, self.n))
        self._f("orc_batch_get_state")(self._at(0), self.N, _dp(x), _dp(P))
        return x, P

    def _get(self, name, width, *args):
        out = np.empty((self.N, width))
        f = self._f(name)
        for i in range(self.N):
            f(self._at(i), *args, _dp(out[i]))
        return out

    def pose(self):
        return self._get("orc_target_get_pose", 7)

    def twist(self):
        return self._get("orc_target_get_twist", 6)

    def acceleration(self):
        return self._get("orc_target_get_acceleration", 6)

    def measured_pose(self):
        return self._get("orc_target_get_measured_pose", 7)

    def pose6(self):
        return self._get("orc_target_get_pose6", 6)

    def transform(self):
        return self._get("orc_target_get_transform", 16).reshape(self.N, 4, 4)

    def period_estimate(self):
        f = self._f("orc_target_get_period_estimate")
        return np.array([f(self._at(i)) for i in range(self.N)])

    def pose_at(self, t1):
        return self._get("orc_target_get_pose_at", 7, float(t1))

    def twist_at(self, t1):
        return self._get("orc_target_get_twist_at", 6, float(t1))

    def acceleration_at(self, t1):
        return self._get("orc_target_get_acceleration_at", 6, float(t1))

    def intersection_time(self, t1, origin, radius):
        origin = _d(origin, (3,))
        f = self._f("orc_intersection_time")
        return np.array([f(self._at(i), float(t1), _dp(origin), float(radius)) for i in range(self.N)])

    def intersection_pose(self, t1, origin, radius):
        origin = _d(origin, (3,))
        f = self._f("orc_intersection_pose")
        pose = np.empty((self.N, 7)); delta = np.empty(self.N); ok = np.empty(self.N, dtype=bool)
        d = C.c_double()
        for i in range(self.N):
            ok[i] = bool(f(self._at(i), float(t1), _dp(origin), float(radius), _dp(pose[i]), C.byref(d)))
            delta[i] = d.value
        return ok, pose, delta


class OracleTarget(OracleBatch):
    """One target (batch of 1) with the reference's per-target call names."""

    def __init__(self, model, Q, R, P0, p0, dt0, t0=0.0, v0=None, a0=None, dtype="f64"):
        super().__init__(model, Q, R, P0, np.asarray(p0).reshape(1, 7), dt0, t0,
                         None if v0 is None else np.asarray(v0).reshape(1, 6),
                         None if a0 is None else np.asarray(a0).reshape(1, 6), dtype)

    def add_measurement(self, dt, meas):
        self.step(dt, np.asarray(meas, dtype=np.float64).reshape(1, 7))

    def update(self, dt):
        self.step(dt, None)


class OracleGate:
    """N independent convergence gates of IntersectionSolver::getIntersectionPoseWithSphere
    (src/intersection_solver.cpp:105-120; one solver object per target)."""

    def __init__(self, n, filters_length=250):
        self.lib = load()
        self.N = n
        self.size = self.lib.orc_gate_sizeof()
        self.buf = (C.c_char * (self.size * n))()
        self.base = C.addressof(self.buf)
        for i in range(n):
            self.lib.orc_gate_init(C.c_void_p(self.base + i * self.size), int(filters_length))

    def update(self, exists, pose, pos_th, ang_th):
        pose = _d(pose, (self.N, 7))
        conv = np.zeros(self.N, dtype=bool)
        pf = np.zeros(self.N); af = np.zeros(self.N)
        a, b = C.c_double(), C.c_double()
        for i in range(self.N):
            conv[i] = bool(self.lib.orc_gate_update(C.c_void_p(self.base + i * self.size), int(bool(exists[i])), _dp(pose[i]),
                                                    float(pos_th), float(ang_th), C.byref(a), C.byref(b)))
            pf[i], af[i] = a.value, b.value
        return conv, pf, af


def ref_test_stream(n_points=10000, dt=1.0 / 250.0, goal=(0.2, 0.3, 0.4), omega=(3.0, 0.01, 0.1),
                    n_models=4):
    """meas[k] = the measurement rows the reference's k-th TEST feeds its filter
    (test/target_manager_test.cpp:82-115, tests in order UV, UA, AR, AV)."""
    lib = load()
    out = np.empty((n_models, n_points, 7))
    lib.orc_ref_test_stream(_dp(out), n_models, n_points, float(dt), _dp(_d(goal)), _dp(_d(omega)))
    return out


def lowest_real_root(coeffs):
    c = _d(coeffs)
    return load().orc_lowest_real_root(_dp(c), len(c))


def poly_roots(coeffs):
    c = _d(coeffs)
    out = np.zeros(2 * (len(c) - 1))
    k = load().orc_poly_roots(_dp(c), len(c), _dp(out))
    return out[:2 * k:2] + 1j * out[1:2 * k:2]


def harness_run(model, Q, R, P0, meas, dt):
    """The reference test's loop for one target, entirely in C: returns (est_pose [n,7], est_twist [n,6])."""
    meas = _d(meas)
    n = meas.shape[0]
    pose = np.empty((n, 7)); twist = np.empty((n, 6))
    load().orc_harness_run_f64(int(model), _dp(_d(Q)), _dp(_d(R)), _dp(_d(P0)), _dp(meas), n, float(dt), _dp(pose), _dp(twist))
    return pose, twist


def stream_fill(model, seed, n_targets, n_ticks, dt, first_target=0, first_tick=0, availability=1.0, rpy_noise=0.0, dtype="f64"):
    """CPU twin of the product's stream generator (oracle/te_stream.c): dict(p0 [N,7], meas [ticks,N,7] -- the
    reference's row layout, values as a ring of precision `dtype` holds them --, has_meas [ticks,N] uint8 or None,
    truth [N,12] = p v a omega).  Bit-identical to target_stream_fill_dev / target_stream_truth_dev."""
    N, T = int(n_targets), int(n_ticks)
    meas = np.empty((T, N, 7)); p0 = np.empty((N, 7)); truth = np.empty((N, 12))
    has = np.empty((T, N), dtype=np.uint8) if availability < 1.0 else None
    load().orc_stream_fill(int(model), int(seed), int(first_target), N, int(first_tick), T, float(dt), float(availability),
                           float(rpy_noise), 1 if dtype == "f32" else 0, _dp(meas),
                           None if has is None else has.ctypes.data_as(C.POINTER(C.c_ubyte)), _dp(p0), _dp(truth))
    return dict(p0=p0, meas=meas, has_meas=has, truth=truth)


def stream_sample(model, seed, targets, n_ticks, dt, first_tick=0, availability=1.0, rpy_noise=0.0, dtype="f64"):
    """The rows of stream_fill for an arbitrary set of targets (e.g. a random sample of a 10^6-target population):
    dict(p0 [k,7], meas [ticks,k,7], has_meas [ticks,k] or None, truth [k,12])."""
    targets = np.asarray(targets, dtype=np.int64)
    k, T = len(targets), int(n_ticks)
    meas = np.empty((T, k, 7)); p0 = np.empty((k, 7)); truth = np.empty((k, 12))
    has = np.empty((T, k), dtype=np.uint8) if availability < 1.0 else None
    one = np.empty((max(T, 1), 7)); h1 = np.empty(max(T, 1), dtype=np.uint8)
    lib = load()
    for j, tg in enumerate(targets):
        lib.orc_stream_fill(int(model), int(seed), int(tg), 1, int(first_tick), T, float(dt), float(availability), float(rpy_noise),
                            1 if dtype == "f32" else 0, _dp(one), h1.ctypes.data_as(C.POINTER(C.c_ubyte)), _dp(p0[j]), _dp(truth[j]))
        meas[:, j] = one[:T]
        if has is not None:
            has[:, j] = h1[:T]
    return dict(p0=p0, meas=meas, has_meas=has, truth=truth)
