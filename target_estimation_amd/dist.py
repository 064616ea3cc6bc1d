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
