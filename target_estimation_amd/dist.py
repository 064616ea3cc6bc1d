"""Multi-GPU plumbing: targets are independent, so the batch shards by contiguous id ranges, one
process (and one TargetManager) per GPU, with no collective on the predict/update path.  The only
exchange the path offers is an optional gather of the estimated poses to one rank
(torch.distributed: backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests)."""
import torch
import torch.distributed as dist


def shard_bounds(n_total, rank, world):
    """Contiguous, balanced id range [lo, hi) of `rank`: the first n_total % world ranks get one
    extra target.  Ascending ids stay ascending across ranks (the reference enumerates its
    std::map in id order, src/target_manager.cpp:126-133)."""
    base, rem = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def owner_of(target_index, n_total, world):
    """Rank that owns global target index `target_index` under shard_bounds."""
    base, rem = divmod(int(n_total), int(world))
    cut = rem * (base + 1)
    if target_index < cut:
        return target_index // (base + 1)
    return rem + (target_index - cut) // max(base, 1)


def gather_rows(local, n_total, dst=0, group=None):
    """Gather per-target rows (e.g. pose7 [n_local, 7]) of every rank into [n_total, w] on `dst`
    (None elsewhere).  Shards may differ by one row, so rows are padded to the largest shard and
    the padding is dropped after the gather.  One message per rank to the root: a direct gather,
    not a ring (xGMI is point-to-point)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    w = local.shape[1]
    max_rows = -(-int(n_total) // world)
    buf = torch.zeros((max_rows, w), dtype=local.dtype, device=local.device)
    buf[: local.shape[0]] = local
    parts = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, parts, dst=dst, group=group)
    if rank != dst:
        return None
    out = []
    for r in range(world):
        lo, hi = shard_bounds(n_total, r, world)
        out.append(parts[r][: hi - lo])
    return torch.cat(out, 0)


def all_gather_rows(local, n_total, group=None):
    """Every rank gets all rows (all_gather_into_tensor on padded shards)."""
    world = dist.get_world_size(group)
    w = local.shape[1]
    max_rows = -(-int(n_total) // world)
    buf = torch.zeros((max_rows, w), dtype=local.dtype, device=local.device)
    buf[: local.shape[0]] = local
    full = torch.empty((world * max_rows, w), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(full, buf, group=group)
    out = []
    for r in range(world):
        lo, hi = shard_bounds(n_total, r, world)
        out.append(full[r * max_rows: r * max_rows + (hi - lo)])
    return torch.cat(out, 0)


class PoseGather:
    """The library's own gather of pose7 rows over xGMI (csrc/pose_gather.cpp: direct ncclSend / ncclRecv on a second
    stream behind an event, overlapped with the following ticks).  torch.distributed is used only to hand rank 0's
    RCCL id and the per-rank row counts to every rank; with world size 1 (or no process group) it is not needed.

        g = PoseGather(mgr)           # collective: every rank
        g.begin()                     # enqueue; returns at once, the next ticks may be issued
        ...                           # mgr steps on
        poses = g.wait()              # root: CUDA double tensor [total, 7] (ranks in order, batch then slot order)
    """

    def __init__(self, manager, root=0, group=None):
        import ctypes as C
        from . import capi
        self._lib = capi.lib()
        self._mgr = manager
        self.root = root
        have_pg = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if have_pg else 0
        self.world = dist.get_world_size(group) if have_pg else 1
        ident = C.create_string_buffer(128)
        if self.rank == 0:
            if self._lib.target_comm_unique_id(ident) != 0:
                raise RuntimeError("target_comm_unique_id failed: %s" % capi.last_error())
        if self.world > 1:
            box = [bytes(ident.raw)]
            dist.broadcast_object_list(box, src=0, group=group)
            ident = C.create_string_buffer(box[0], 128)
        self._h = self._lib.target_comm_new(ident, self.rank, self.world)
        if not self._h:
            raise RuntimeError("target_comm_new failed: %s" % capi.last_error())
        self._group = group
        self._recv = None
        self._counts = None

    def counts(self):
        """rows per rank (exchanged once per call of begin(): targets may have been created or erased)"""
        mine = int(self._mgr.size())
        if self.world == 1:
            return [mine]
        box = [None] * self.world
        dist.all_gather_object(box, mine, group=self._group)
        return [int(v) for v in box]

    def begin(self, counts=None):
        import ctypes as C
        from . import capi
        counts = self.counts() if counts is None else list(counts)
        total = sum(counts)
        arr = (C.c_long * self.world)(*counts)
        ptr = None
        if self.rank == self.root:
            if self._recv is None or self._recv.shape[0] != total:
                self._recv = torch.empty((total, 7), dtype=torch.float64, device="cuda")
            ptr = self._recv.data_ptr()
        if self._lib.target_manager_gather_pose_begin(self._mgr.handle, self._h, int(self.root), arr, ptr) != 0:
            raise RuntimeError("target_manager_gather_pose_begin failed: %s" % capi.last_error())
        self._counts = counts

    def wait(self):
        """Blocks until the gather has finished; returns (poses or None, device milliseconds of the gather)."""
        import ctypes as C
        from . import capi
        ms = C.c_float()
        deadline = getattr(self, "deadline", None)   # time.monotonic() value: bounded wait (a peer may never arrive)
        if deadline is None:
            if self._lib.target_manager_gather_pose_wait(self._h, C.byref(ms)) != 0:
                raise RuntimeError("target_manager_gather_pose_wait failed: %s" % capi.last_error())
        else:
            import time
            rc = self._lib.target_manager_gather_pose_wait_for(self._h, max(0.0, deadline - time.monotonic()), C.byref(ms))
            if rc < 0:
                raise RuntimeError("target_manager_gather_pose_wait_for failed: %s" % capi.last_error())
            if rc == 1:
                raise TimeoutError("pose gather still in flight at its deadline (rank %d of %d)" % (self.rank, self.world))
        return (self._recv if self.rank == self.root else None), float(ms.value)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.target_comm_delete(self._h)
            self._h = None

    __del__ = close
