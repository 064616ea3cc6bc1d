"""Python front-end over the C ABI (ctypes).  Mirrors the reference's TargetManager method names
(include/target_estimation/target_manager.hpp:66-203); every numeric call goes through
libtarget_estimation_amd.so and runs on the GPU.  torch is used only for device buffers/streams.
"""
import ctypes as C

import numpy as np

from . import capi

ANGULAR_RATES, ANGULAR_VELOCITIES, UNIFORM_ACCELERATION, UNIFORM_VELOCITY = 0, 1, 2, 3
MODEL_TYPES = {"angular_rates": 0, "angular_velocities": 1, "uniform_acceleration": 2, "uniform_velocity": 3}
MODEL_DIMS = {0: (18, 6), 1: (12, 6), 2: (9, 3), 3: (6, 3)}
DTYPES = {"f64": 0, "f32": 1}
SYMMETRIC_PACKED = 100   # add to lanes_per_target=1: upper triangle of P only in HBM
AXIS_SEPARABLE = 200     # add to lanes_per_target=1: only the per-axis blocks of P (exact for decoupled Q, R, P0)


def _dp(a):
    return None if a is None else a.ctypes.data_as(capi.c_double_p)


def _d(a, shape=None):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _ids(ids):
    ids = np.ascontiguousarray(ids, dtype=np.uint32)
    return ids, ids.ctypes.data_as(capi.c_uint_p)


def _check(rc, what):
    if rc is None or rc < 0:
        raise RuntimeError("%s failed: %s" % (what, capi.last_error()))
    return rc


class Batch:
    """All targets of one (model, Q, R) of a manager: the device-resident dense path."""

    def __init__(self, lib, handle):
        self._lib, self._h = lib, handle

    size = property(lambda s: s._lib.target_batch_size(s._h))
    type = property(lambda s: s._lib.target_batch_type(s._h))
    dtype = property(lambda s: "f64" if s._lib.target_batch_dtype(s._h) == 0 else "f32")
    state_dim = property(lambda s: s._lib.target_batch_state_dim(s._h))
    meas_dim = property(lambda s: s._lib.target_batch_meas_dim(s._h))
    lanes_per_target = property(lambda s: s._lib.target_batch_lanes_per_target(s._h))
    symmetric_packed = property(lambda s: bool(s._lib.target_batch_is_symmetric_packed(s._h)))
    layout = property(lambda s: ("full", "symmetric_packed", "axis_separable", "axis_separable_packed")[s._lib.target_batch_layout(s._h)])
    num_classes = property(lambda s: s._lib.target_batch_num_classes(s._h))
    algorithmic_bytes = property(lambda s: s._lib.target_batch_algorithmic_bytes(s._h))
    resident_bytes_per_target = property(lambda s: s._lib.target_batch_resident_bytes_per_target(s._h))

    def slot_ids(self):
        n = self.size
        out = np.empty(n, dtype=np.uint32)
        self._lib.target_batch_slot_ids(self._h, out.ctypes.data_as(capi.c_uint_p), n)
        return out

    def torch_dtype(self):
        import torch
        return torch.float64 if self.dtype == "f64" else torch.float32

    def step(self, dt, meas=None, has_meas=None):
        """One tick over every target.  meas: CUDA tensor [7, ld] (SoA, batch precision) or None
        (predict only); has_meas: CUDA uint8 tensor [size] or None.  Asynchronous."""
        mp, ld, hp = None, 0, None
        if meas is not None:
            assert meas.is_cuda and meas.dim() == 2 and meas.shape[0] == 7 and (meas.shape[1] == 1 or meas.stride(1) == 1)
            assert meas.dtype == self.torch_dtype(), (meas.dtype, self.dtype)
            assert meas.shape[1] >= self.size
            mp, ld = meas.data_ptr(), meas.stride(0)
        if has_meas is not None:
            assert has_meas.is_cuda and has_meas.numel() >= self.size and has_meas.element_size() == 1
            hp = has_meas.data_ptr()
        _check(self._lib.target_batch_step(self._h, float(dt), mp, ld, hp), "target_batch_step")

    def step_host(self, dt, meas_soa, has_meas=None):
        """One tick from HOST measurements: meas_soa = CPU tensor or ndarray [rows >= 3 or 7, ld] in the batch
        precision (SoA; rows >= 7 for the angular models; pinned memory gives asynchronous DMA), has_meas = CPU uint8 [size] or None."""
        import torch
        t = meas_soa if isinstance(meas_soa, torch.Tensor) else torch.from_numpy(meas_soa)
        assert not t.is_cuda and t.dim() == 2 and t.stride(1) == 1 and t.dtype == self.torch_dtype() and t.shape[1] >= self.size
        need = 7 if self.type in (ANGULAR_RATES, ANGULAR_VELOCITIES) else 3   # rows the model reads (and the copy moves)
        assert t.shape[0] >= need, "this model reads %d measurement rows, got %d" % (need, t.shape[0])
        hp = None
        if has_meas is not None:
            h = has_meas if isinstance(has_meas, torch.Tensor) else torch.from_numpy(has_meas)
            assert not h.is_cuda and h.element_size() == 1 and h.numel() >= self.size
            hp = h.data_ptr()
        _check(self._lib.target_batch_step_host(self._h, float(dt), t.data_ptr(), t.stride(0), hp), "target_batch_step_host")

    def step_sequence(self, dt, meas, has_meas=None, use_graph=False, n_ticks=None):
        """meas: CUDA tensor [ticks, 7, ld]: one launch per tick, all enqueued by one C call.  n_ticks > ticks
        treats meas (and has_meas) as a ring: tick s reads entry s % ticks."""
        assert meas.is_cuda and meas.dim() == 3 and meas.shape[1] == 7 and (meas.shape[2] == 1 or meas.stride(2) == 1)
        assert meas.dtype == self.torch_dtype() and meas.shape[2] >= self.size
        hp, hs = None, 0
        if has_meas is not None:
            assert has_meas.is_cuda and has_meas.dim() == 2 and has_meas.element_size() == 1
            hp, hs = has_meas.data_ptr(), has_meas.stride(0)
        if n_ticks is None or n_ticks == meas.shape[0]:
            _check(self._lib.target_batch_step_sequence(self._h, meas.shape[0], float(dt), meas.data_ptr(), meas.stride(0),
                                                         meas.stride(1), hp, hs, int(use_graph)),
                   "target_batch_step_sequence")
        else:
            _check(self._lib.target_batch_step_sequence_ring(self._h, int(n_ticks), float(dt), meas.data_ptr(), meas.stride(0),
                                                              meas.stride(1), hp, hs, meas.shape[0], int(use_graph)),
                   "target_batch_step_sequence_ring")

    def step_fused(self, dt, meas, has_meas=None):
        """meas: CUDA tensor [ticks, 7, ld]: all ticks in ONE launch (state stays in registers)."""
        assert meas.is_cuda and meas.dim() == 3 and meas.shape[1] == 7 and (meas.shape[2] == 1 or meas.stride(2) == 1)
        assert meas.dtype == self.torch_dtype() and meas.shape[2] >= self.size
        hp, hs = None, 0
        if has_meas is not None:
            assert has_meas.is_cuda and has_meas.dim() == 2 and has_meas.element_size() == 1
            hp, hs = has_meas.data_ptr(), has_meas.stride(0)
        _check(self._lib.target_batch_step_fused(self._h, meas.shape[0], float(dt), meas.data_ptr(), meas.stride(0),
                                                  meas.stride(1), hp, hs), "target_batch_step_fused")

    # ---- resident ("live") mode: one launch serves tick after tick as the host posts them (target_batch_c.h)
    def live_start(self, dt, meas_ring, has_ring=None, first_entry=0, max_ticks=1 << 30, idle_limit_s=10.0):
        """meas_ring: CUDA tensor [ring_ticks, 7, ld] in the batch precision; has_ring: CUDA uint8 [ring_ticks, >= size] or None."""
        assert meas_ring.is_cuda and meas_ring.dim() == 3 and meas_ring.shape[1] == 7 and meas_ring.stride(2) == 1
        assert meas_ring.dtype == self.torch_dtype() and meas_ring.shape[2] >= self.size
        hp, hs = None, 0
        if has_ring is not None:
            assert has_ring.is_cuda and has_ring.dim() == 2 and has_ring.element_size() == 1 and has_ring.shape[0] == meas_ring.shape[0]
            hp, hs = has_ring.data_ptr(), has_ring.stride(0)
        _check(self._lib.target_batch_live_start(self._h, float(dt), meas_ring.data_ptr(), meas_ring.stride(0), meas_ring.stride(1), hp, hs,
                                                 meas_ring.shape[0], int(first_entry), int(max_ticks), float(idle_limit_s)), "target_batch_live_start")

    def live_set_pose_output(self, pose_soa):
        """pose_soa: CUDA double tensor [7, ld >= size] (or None): sessions started afterwards write every tick's poses there."""
        if pose_soa is None:
            _check(self._lib.target_batch_live_set_pose_output(self._h, None, 0), "target_batch_live_set_pose_output")
            return
        import torch
        assert pose_soa.is_cuda and pose_soa.dtype == torch.float64 and pose_soa.dim() == 2 and pose_soa.shape[0] == 7 and pose_soa.stride(1) == 1
        _check(self._lib.target_batch_live_set_pose_output(self._h, pose_soa.data_ptr(), pose_soa.stride(0)), "target_batch_live_set_pose_output")

    def live_post(self, n_ticks=1):
        _check(self._lib.target_batch_live_post(self._h, int(n_ticks)), "target_batch_live_post")

    def live_post_each(self, n_ticks):
        _check(self._lib.target_batch_live_post_each(self._h, int(n_ticks)), "target_batch_live_post_each")

    def live_done(self):
        return self._lib.target_batch_live_done(self._h)

    def live_wait(self, tick, timeout_s=5.0):
        rc = _check(self._lib.target_batch_live_wait(self._h, int(tick), float(timeout_s)), "target_batch_live_wait")
        return rc == 0

    def live_stop(self):
        return _check(self._lib.target_batch_live_stop(self._h), "target_batch_live_stop")

    live_capacity = property(lambda s: s._lib.target_batch_live_capacity(s._h))

    def live_running(self):
        """True while the session's resident kernel is there (it leaves after live_stop, or by itself after the idle limit)."""
        return self._lib.target_batch_live_running(self._h) == 1

    def get_est(self, pose=True, twist=True, acc=True, t1=None):
        """Derived outputs of every slot as CUDA double tensors ([size,7], [size,6], [size,6])."""
        import torch
        n = self.size
        mk = lambda w, on: torch.empty((n, w), dtype=torch.float64, device="cuda") if on else None  # noqa: E731
        p, t, a = mk(7, pose), mk(6, twist), mk(6, acc)
        ptr = lambda x: None if x is None else x.data_ptr()  # noqa: E731
        _check(self._lib.target_batch_get_est_dev(self._h, ptr(p), ptr(t), ptr(a), 0 if t1 is None else 1,
                                                   0.0 if t1 is None else float(t1)), "target_batch_get_est_dev")
        return p, t, a

    def intersect_sphere(self, origin, radius, t1=None, want_pose=True):
        """Sphere-intersection query for every slot: (delta [size], pose [size,7] or None) as CUDA double
        tensors; t1=None queries each target at its own current time."""
        import torch
        n = self.size
        delta = torch.empty(n, dtype=torch.float64, device="cuda")
        pose = torch.empty((n, 7), dtype=torch.float64, device="cuda") if want_pose else None
        origin = _d(origin, (3,))
        _check(self._lib.target_batch_intersect_sphere_dev(
            self._h, float("nan") if t1 is None else float(t1), _dp(origin), float(radius), delta.data_ptr(),
            None if pose is None else pose.data_ptr()), "target_batch_intersect_sphere_dev")
        return delta, pose

    def intersect_sphere_converged(self, origin, radius, pos_th, ang_th, t1=None, filters_length=250):
        import torch
        n = self.size
        delta = torch.empty(n, dtype=torch.float64, device="cuda")
        pose = torch.empty((n, 7), dtype=torch.float64, device="cuda")
        conv = torch.empty(n, dtype=torch.uint8, device="cuda")
        origin = _d(origin, (3,))
        _check(self._lib.target_batch_intersect_sphere_converged_dev(
            self._h, float("nan") if t1 is None else float(t1), float(pos_th), float(ang_th), _dp(origin), float(radius),
            int(filters_length), delta.data_ptr(), pose.data_ptr(), conv.data_ptr()), "target_batch_intersect_sphere_converged_dev")
        return conv, pose, delta

    def gate_update(self, delta, pose, pos_th, ang_th, filters_length=250, want_variance=False):
        """The convergence gate alone on query results already on the device (delta [size], pose [size,7], CUDA doubles):
        returns converged [size] uint8, filtered errors [size,2] and, on request, the filters' variances [size,2]."""
        import torch
        n = self.size
        assert delta.is_cuda and pose.is_cuda and delta.dtype == torch.float64 and pose.dtype == torch.float64
        assert delta.numel() >= n and pose.is_contiguous() and pose.shape[0] >= n and pose.shape[1] == 7
        conv = torch.empty(n, dtype=torch.uint8, device="cuda")
        filt = torch.empty((n, 2), dtype=torch.float64, device="cuda")
        var = torch.empty((n, 2), dtype=torch.float64, device="cuda") if want_variance else None
        _check(self._lib.target_batch_gate_update_dev(self._h, delta.data_ptr(), pose.data_ptr(), float(pos_th), float(ang_th),
                                                       int(filters_length), conv.data_ptr(), filt.data_ptr(),
                                                       None if var is None else var.data_ptr()), "target_batch_gate_update_dev")
        return conv, filt, var

    def pack_meas(self, meas_aos, out=None):
        """CUDA double [n,7] (the reference's row layout) -> SoA [7,n] in the batch precision."""
        import torch
        n = meas_aos.shape[0]
        assert meas_aos.is_cuda and meas_aos.dtype == torch.float64 and meas_aos.is_contiguous()
        if out is None:
            out = torch.empty((7, n), dtype=self.torch_dtype(), device="cuda")
        _check(self._lib.target_batch_pack_meas_dev(self._h, meas_aos.data_ptr(), n, out.data_ptr(), out.stride(0)),
               "target_batch_pack_meas_dev")
        return out


class MeasurementIngest:
    """The ROS node's mailbox / has-measurement / expiry policy (RosTargetManager::update) over a manager."""

    def __init__(self, manager, type=None, Q=None, R=None, P0=None, expiration_time=None, token=None):
        self._lib = manager._lib
        self._mgr = manager
        if Q is None:
            self._h = self._lib.target_ingest_new(manager.handle, 0, None, None, None)
        else:
            n, m = MODEL_DIMS[int(type)]
            self._h = self._lib.target_ingest_new(manager.handle, int(type), _dp(_d(Q, (n, n))), _dp(_d(R, (m, m))), _dp(_d(P0, (n, n))))
        if not self._h:
            raise RuntimeError("target_ingest_new failed: %s" % capi.last_error())
        if expiration_time is not None:
            self._lib.target_ingest_set_expiration_time(self._h, float(expiration_time))
        if token is not None:
            self._lib.target_ingest_set_token_name(self._h, token.encode())

    def close(self):
        if getattr(self, "_h", None):
            self._lib.target_ingest_delete(self._h)
            self._h = None

    def push(self, id, stamp, pose):
        _check(self._lib.target_ingest_push(self._h, int(id), float(stamp), _dp(_d(pose, (7,)))), "target_ingest_push")

    def push_named(self, frame, stamp, pose):
        return self._lib.target_ingest_push_named(self._h, frame.encode(), float(stamp), _dp(_d(pose, (7,))))

    def tick(self, dt, now, capacity=4096):
        ids = np.zeros(capacity, dtype=np.uint32)
        poses = np.zeros((capacity, 7))
        n = _check(self._lib.target_ingest_tick(self._h, float(dt), float(now), ids.ctypes.data_as(capi.c_uint_p), _dp(poses),
                                                capacity), "target_ingest_tick")
        return ids[:n].copy(), poses[:n].copy()


class TargetManager:
    """ctypes mirror of the reference TargetManager; dtype 'f64' (reference precision) or 'f32'."""

    def __init__(self, file=None, dtype="f64", lanes_per_target=0):
        self._lib = capi.lib()
        f = None if file is None else str(file).encode()
        self._h = self._lib.target_manager_new_ex(f, DTYPES[dtype], int(lanes_per_target))
        if not self._h:
            raise RuntimeError("target_manager_new failed: %s" % capi.last_error())
        self.dtype = dtype

    def close(self):
        if getattr(self, "_h", None):
            self._lib.target_manager_delete(self._h)
            self._h = None

    __del__ = close

    @property
    def handle(self):
        return self._h

    # ---- the reference's per-target calls ---------------------------------------------------
    def init(self, id, dt0, t0, p0, v0=None, a0=None, type=None, Q=None, R=None, P0=None):
        p0 = _d(p0, (7,))
        if type is None:
            assert v0 is None and a0 is None, "the C boundary's init takes a pose only (target_manager_c.h:29)"
            self._lib.target_manager_init(self._h, int(id), float(dt0), _dp(p0), float(t0))
            return
        n, m = MODEL_DIMS[int(type)]
        Q, R, P0 = _d(Q, (n, n)), _d(R, (m, m)), _d(P0, (n, n))
        v0, a0 = _d(v0, (6,)), _d(a0, (6,))
        _check(self._lib.target_manager_init_typed(self._h, int(type), int(id), float(dt0), float(t0), _dp(Q), _dp(R),
                                                    _dp(P0), _dp(p0), _dp(v0), _dp(a0)), "target_manager_init_typed")

    def update(self, id, dt, meas=None):
        if meas is None:
            self._lib.target_manager_update(self._h, int(id), float(dt))
        else:
            meas = _d(meas, (7,))
            self._lib.target_manager_update_meas(self._h, int(id), float(dt), _dp(meas))

    def update_all(self, dt):
        _check(self._lib.target_manager_update_all(self._h, float(dt)), "target_manager_update_all")

    def erase(self, id):
        return bool(_check(self._lib.target_manager_erase(self._h, int(id)), "target_manager_erase"))

    def erase_batch(self, ids):
        ids, idp = _ids(ids)
        return self._lib.target_manager_erase_batch(self._h, idp, len(ids))

    def _get1(self, fn, id, w):
        out = np.full(w, np.nan)
        ok = fn(self._h, int(id), _dp(out))
        return bool(ok), out

    def getTargetPose(self, id):
        return self._get1(self._lib.target_manager_get_est_pose, id, 7)

    def getTargetTwist(self, id):
        return self._get1(self._lib.target_manager_get_est_twist, id, 6)

    def getTargetAcceleration(self, id):
        return self._get1(self._lib.target_manager_get_est_acceleration, id, 6)

    def getNumberMeasurements(self, id):
        return self._lib.target_manager_get_n_measurements(self._h, int(id))

    def getAvailableTargets(self):
        n = self._lib.target_manager_size(self._h)
        out = np.empty(max(n, 1), dtype=np.uint32)
        n = self._lib.target_manager_get_available_targets(self._h, out.ctypes.data_as(capi.c_uint_p), n)
        return out[:n]

    def getTime(self, id):
        t = C.c_double()
        rc = self._lib.target_manager_get_time(self._h, int(id), C.byref(t))
        return t.value if rc == 0 else None

    def log(self):
        self._lib.target_manager_log(self._h)

    def set_log_directory(self, path):
        _check(self._lib.target_manager_set_log_directory(self._h, None if path is None else str(path).encode()),
               "target_manager_set_log_directory")

    def set_log_targets(self, ids):
        ids, idp = _ids(ids if ids is not None else [])
        _check(self._lib.target_manager_set_log_targets(self._h, idp, len(ids)), "target_manager_set_log_targets")

    def set_keep_measurement(self, on=True):
        _check(self._lib.target_manager_set_keep_measurement(self._h, 1 if on else 0), "target_manager_set_keep_measurement")

    # TargetInterface getters reached through getTarget(id)-> in the reference (target_interface.hpp:94-148)
    def getMeasuredPose(self, id):
        return self._get1(self._lib.target_manager_get_measured_pose, id, 7)

    def getPeriodEstimate(self, id):
        p = C.c_double(float("nan"))
        ok = self._lib.target_manager_get_period_estimate(self._h, int(id), C.byref(p))
        return p.value if ok else None

    def getEstimatedTransform(self, id):
        ok, T = self._get1(self._lib.target_manager_get_estimated_transform, id, 16)
        return bool(ok), T.reshape(4, 4)

    def getN(self, id):
        return self._lib.target_manager_get_n(self._h, int(id))

    def getM(self, id):
        return self._lib.target_manager_get_m(self._h, int(id))

    def getModelMatrices(self, id):
        """(Q, R, P0) the target was created with (getEstimator()->getQ / getR / getP0), or None for an unknown id."""
        n, m = self.getN(id), self.getM(id)
        if n <= 0:
            return None
        Q, R, P0 = np.empty((n, n)), np.empty((m, m)), np.empty((n, n))
        ok = self._lib.target_manager_get_model_matrices(self._h, int(id), _dp(Q), _dp(R), _dp(P0))
        return (Q, R, P0) if ok else None

    def size(self):
        return self._lib.target_manager_size(self._h)

    def synchronize(self):
        _check(self._lib.target_manager_synchronize(self._h), "target_manager_synchronize")

    def set_stream(self, stream_ptr):
        _check(self._lib.target_manager_set_stream(self._h, stream_ptr), "target_manager_set_stream")

    # ---- batched extension ---------------------------------------------------------------------
    def init_batch(self, ids, dt0, t0, p0, v0=None, a0=None, type=None, Q=None, R=None, P0=None):
        ids, idp = _ids(ids)
        n = len(ids)
        p0, v0, a0 = _d(p0, (n, 7)), _d(v0, (n, 6)), _d(a0, (n, 6))
        if type is None:
            return _check(self._lib.target_manager_init_batch(self._h, idp, n, float(dt0), float(t0), _dp(p0), _dp(v0),
                                                              _dp(a0)), "target_manager_init_batch")
        ns, m = MODEL_DIMS[int(type)]
        Q, R = _d(Q, (ns, ns)), _d(R, (m, m))
        P0 = _d(P0)
        per = 1 if P0.ndim == 3 else 0
        return _check(self._lib.target_manager_init_batch_typed(
            self._h, int(type), idp, n, float(dt0), float(t0), _dp(Q), _dp(R), _dp(np.ascontiguousarray(P0)), per,
            _dp(p0), _dp(v0), _dp(a0)), "target_manager_init_batch_typed")

    def init_batch_classes(self, ids, dt0, t0, p0, type, Q, R, P0, class_of, v0=None, a0=None):
        """n targets with per-class parameters: Q [n_classes, ns, ns], R [n_classes, m, m], P0 [n_classes, ns, ns],
        class_of [n] (the class of every target).  All classes of one layout share one batch (one launch per tick)."""
        ids, idp = _ids(ids)
        n = len(ids)
        ns, m = MODEL_DIMS[int(type)]
        Q, R, P0 = _d(Q), _d(R), _d(P0)
        nc = Q.shape[0]
        assert Q.shape == (nc, ns, ns) and R.shape == (nc, m, m) and P0.shape == (nc, ns, ns)
        class_of = np.ascontiguousarray(class_of, dtype=np.uint32)
        assert class_of.shape == (n,) and (class_of < nc).all()
        p0, v0, a0 = _d(p0, (n, 7)), _d(v0, (n, 6)), _d(a0, (n, 6))
        return _check(self._lib.target_manager_init_batch_classes(
            self._h, int(type), idp, n, float(dt0), float(t0), nc, _dp(Q), _dp(R), _dp(P0), class_of.ctypes.data_as(capi.c_uint_p),
            _dp(p0), _dp(v0), _dp(a0)), "target_manager_init_batch_classes")

    def update_batch(self, ids, dt, meas=None, has_meas=None):
        ids, idp = _ids(ids)
        n = len(ids)
        meas = _d(meas, (n, 7))
        hp = None
        if has_meas is not None:
            has_meas = np.ascontiguousarray(has_meas, dtype=np.uint8)
            hp = has_meas.ctypes.data_as(capi.c_ubyte_p)
        return _check(self._lib.target_manager_update_meas_batch(self._h, idp, n, float(dt), _dp(meas), hp),
                      "target_manager_update_meas_batch")

    def get_est_batch(self, ids, t1=None):
        ids, idp = _ids(ids)
        n = len(ids)
        pose, twist, acc = np.full((n, 7), np.nan), np.full((n, 6), np.nan), np.full((n, 6), np.nan)
        found = np.zeros(n, dtype=np.uint8)
        fp = found.ctypes.data_as(capi.c_ubyte_p)
        if t1 is None:
            _check(self._lib.target_manager_get_est_batch(self._h, idp, n, _dp(pose), _dp(twist), _dp(acc), fp),
                   "target_manager_get_est_batch")
        else:
            _check(self._lib.target_manager_get_est_at_batch(self._h, idp, n, float(t1), _dp(pose), _dp(twist), _dp(acc),
                                                             fp), "target_manager_get_est_at_batch")
        return pose, twist, acc, found.astype(bool)

    def get_state_batch(self, ids):
        ids, idp = _ids(ids)
        n = len(ids)
        x = np.empty((n, 18)); P = np.empty((n, 18 * 18))
        ns = _check(self._lib.target_manager_get_state_batch(self._h, idp, n, _dp(x), _dp(P)),
                    "target_manager_get_state_batch")
        return (x.reshape(-1)[:n * ns].reshape(n, ns).copy(),
                P.reshape(-1)[:n * ns * ns].reshape(n, ns, ns).copy())

    def intersection_time(self, id, t1, origin, radius):
        origin = _d(origin, (3,))
        return self._lib.target_manager_get_intersection_time_with_sphere(self._h, int(id), float(t1), _dp(origin), float(radius))

    def intersection_pose(self, id, t1, origin, radius):
        origin = _d(origin, (3,))
        pose = np.zeros(7)
        d = C.c_double()
        ok = self._lib.target_manager_get_intersection_pose_with_sphere(self._h, int(id), float(t1), _dp(origin), float(radius),
                                                                         _dp(pose), C.byref(d))
        return bool(ok), pose, d.value

    def intersect_batch(self, ids, t1, origin, radius):
        ids, idp = _ids(ids)
        n = len(ids)
        origin = _d(origin, (3,))
        delta = np.empty(n); pose = np.empty((n, 7)); found = np.zeros(n, dtype=np.uint8)
        _check(self._lib.target_manager_intersect_sphere_batch(self._h, idp, n, float(t1), _dp(origin), float(radius),
                                                               _dp(delta), _dp(pose), found.ctypes.data_as(capi.c_ubyte_p)),
               "target_manager_intersect_sphere_batch")
        return delta, pose, found.astype(bool)

    def intersect_converged_batch(self, ids, t1, pos_th, ang_th, origin, radius, filters_length=250):
        """IntersectionSolver::getIntersectionPoseWithSphere incl. its convergence gate (one gate per target):
        returns converged [n] bool, pose [n,7], delta [n], filtered errors [n,2]."""
        ids, idp = _ids(ids)
        n = len(ids)
        origin = _d(origin, (3,))
        delta = np.empty(n); pose = np.empty((n, 7)); filt = np.empty((n, 2))
        conv = np.zeros(n, dtype=np.uint8); found = np.zeros(n, dtype=np.uint8)
        _check(self._lib.target_manager_intersect_sphere_converged_batch(
            self._h, idp, n, float(t1), float(pos_th), float(ang_th), _dp(origin), float(radius), int(filters_length),
            _dp(delta), _dp(pose), conv.ctypes.data_as(capi.c_ubyte_p), found.ctypes.data_as(capi.c_ubyte_p), _dp(filt)),
            "target_manager_intersect_sphere_converged_batch")
        return conv.astype(bool), pose, delta, filt

    def step_sequence_all(self, dt, meas, has_meas=None, query=None, use_graph=True, n_ticks=None):
        """meas: one CUDA tensor [ticks, 7, ld] per batch (batches() order): `ticks` ticks of every batch -- ONE launch per
        tick for all of them where population_tick() holds, otherwise a launch per batch (recorded: one graph branch per
        batch) -- replayed from a recorded hipGraph (use_graph) or eagerly.  query =
        (origin[3], radius, deltas, poses) adds the own-time sphere query of every target after every step;
        deltas[i] [size] and poses[i] [size, 7] (or None) are CUDA double tensors, overwritten every tick."""
        nb = len(meas)          # the library checks it against the number of batches
        ring = meas[0].shape[0] if nb else 0
        ticks = ring if n_ticks is None else int(n_ticks)      # n_ticks > ring: the tensors are rings (tick s reads s % ring)
        specs = (capi.BatchSequence * max(nb, 1))()
        for i, t in enumerate(meas):
            assert t.is_cuda and t.dim() == 3 and t.shape[0] == ring and t.shape[1] == 7 and (t.shape[2] == 1 or t.stride(2) == 1)
            specs[i].meas_dev, specs[i].tick_stride, specs[i].ld = t.data_ptr(), t.stride(0), t.stride(1)
            specs[i].ring_ticks = ring if ticks != ring else 0
            if has_meas is not None and has_meas[i] is not None:
                h = has_meas[i]
                assert h.is_cuda and h.dim() == 2 and h.element_size() == 1 and h.shape[0] == ring
                specs[i].has_meas_dev, specs[i].has_stride = h.data_ptr(), h.stride(0)
        origin, radius = None, 0.0
        if query is not None:
            origin, radius, deltas, poses = query
            origin = _d(origin, (3,))
            for i in range(nb):
                specs[i].delta_dev = deltas[i].data_ptr()
                specs[i].pose_dev = None if poses is None or poses[i] is None else poses[i].data_ptr()
        _check(self._lib.target_manager_step_sequence_all(
            self._h, ticks, float(dt), C.cast(specs, C.c_void_p), nb, 0 if query is None else 1,
            None if origin is None else _dp(origin), float(radius), int(use_graph)), "target_manager_step_sequence_all")

    def population_tick(self):
        """True if step_sequence_all steps every batch with ONE launch per tick (target_manager_population_tick)."""
        return _check(self._lib.target_manager_population_tick(self._h), "target_manager_population_tick") == 1

    # ---- resident ("live") mode of every batch at once (target_batch_c.h)
    def live_start_all(self, dt, meas, has_meas=None, first_entry=0, max_ticks=1 << 30, idle_limit_s=10.0, query=None):
        """meas: one CUDA ring tensor [ring_ticks, 7, ld] per batch (batches() order).  query = (origin[3], radius, deltas, poses)
        adds the own-time sphere query of every target after every tick (as step_sequence_all's)."""
        nb = len(meas)
        specs = (capi.BatchSequence * max(nb, 1))()
        for i, t in enumerate(meas):
            assert t.is_cuda and t.dim() == 3 and t.shape[1] == 7 and t.stride(2) == 1
            specs[i].meas_dev, specs[i].tick_stride, specs[i].ld, specs[i].ring_ticks = t.data_ptr(), t.stride(0), t.stride(1), t.shape[0]
            if has_meas is not None and has_meas[i] is not None:
                h = has_meas[i]
                assert h.is_cuda and h.dim() == 2 and h.element_size() == 1 and h.shape[0] == t.shape[0]
                specs[i].has_meas_dev, specs[i].has_stride = h.data_ptr(), h.stride(0)
        origin, radius = None, 0.0
        if query is not None:
            origin, radius, deltas, poses = query
            origin = _d(origin, (3,))
            for i in range(nb):
                specs[i].delta_dev = deltas[i].data_ptr()
                specs[i].pose_dev = None if poses is None or poses[i] is None else poses[i].data_ptr()
        _check(self._lib.target_manager_live_start_all(self._h, float(dt), C.cast(specs, C.c_void_p), nb, int(first_entry), int(max_ticks),
                                                       float(idle_limit_s), 0 if query is None else 1, None if origin is None else _dp(origin),
                                                       float(radius)), "target_manager_live_start_all")

    def live_post_all(self, n_ticks=1, one_doorbell_per_tick=False):
        _check(self._lib.target_manager_live_post_all(self._h, int(n_ticks), 1 if one_doorbell_per_tick else 0), "target_manager_live_post_all")

    def live_done_all(self):
        return self._lib.target_manager_live_done_all(self._h)

    def live_wait_all(self, tick, timeout_s=5.0):
        return _check(self._lib.target_manager_live_wait_all(self._h, int(tick), float(timeout_s)), "target_manager_live_wait_all") == 0

    def live_stop_all(self):
        return _check(self._lib.target_manager_live_stop_all(self._h), "target_manager_live_stop_all")

    def batches(self):
        return [Batch(self._lib, self._lib.target_manager_get_batch(self._h, i))
                for i in range(self._lib.target_manager_num_batches(self._h))]

    def batch_of_type(self, type):
        h = self._lib.target_manager_get_batch_of_type(self._h, int(type))
        return Batch(self._lib, h) if h else None
