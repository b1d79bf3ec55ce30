"""Multi-GPU sharding of a batch of independent gates: keys are replicated per GPU, the
gate stream is cut into contiguous shards balanced by blind-rotation count (MUX = 2, NOT/CONST/COPY = 0),
every rank runs its shard with no communication, and the results are gathered once (RCCL all_gather on
GPUs, gloo on CPU in the tests).  One process per GPU; `torch.distributed` is only the transport."""
import numpy as np

from ._lib import OPCODES

_ROT_COST = np.ones(256, np.int64)
_ROT_COST[OPCODES["MUX"]] = 2
for _name in ("NOT", "CONST0", "CONST1", "COPY"):
    _ROT_COST[OPCODES[_name]] = 0


def rotation_cost(opcodes):
    """Blind rotations each gate costs (gates.jl: MUX runs bootstrap_wo_keyswitch twice, :167,171)."""
    return _ROT_COST[np.asarray(opcodes, np.uint8)]


def shard_bounds(opcodes, world):
    """Contiguous [start, end) per rank, balancing the cumulative rotation count.  Every gate lands in
    exactly one shard; shards may be empty when there are fewer gates than ranks.  Exact integer arithmetic,
    the same rule as the library's tfhe_shard_bounds (a multi-device context shards a batch this way itself):
    a gate weighs 1000 x its rotations + 1, shard r starts at the first gate g where
    weight(gates before g) x world >= total weight x r."""
    ops = np.asarray(opcodes, np.uint8)
    B = ops.size
    if B == 0:
        return [(0, 0)] * world
    cost = 1000 * rotation_cost(ops) + 1
    cum = np.concatenate([[0], np.cumsum(cost)]).astype(np.int64)
    cuts = [0]
    for r in range(1, world):
        cuts.append(int(np.searchsorted(cum * world, cum[-1] * r, side="left")))
    cuts.append(B)
    cuts = np.maximum.accumulate(np.minimum(cuts, B))
    return [(int(cuts[r]), int(cuts[r + 1])) for r in range(world)]


def gather_shards(local_out, bounds, rank, group=None):
    """All ranks receive the full [B][width] result.  `local_out`: torch tensor [end-start][width] on the
    rank's device.  Uneven shards are padded to the longest one for the single all_gather."""
    import torch
    import torch.distributed as dist

    world = len(bounds)
    width = local_out.shape[1]
    longest = max(e - s for s, e in bounds)
    padded = torch.zeros((longest, width), dtype=local_out.dtype, device=local_out.device)
    padded[: local_out.shape[0]] = local_out
    full = torch.empty((world * longest, width), dtype=local_out.dtype, device=local_out.device)
    dist.all_gather_into_tensor(full, padded, group=group)
    parts = [full[r * longest: r * longest + (e - s)] for r, (s, e) in enumerate(bounds)]
    return torch.cat(parts, dim=0)


def gather_to_root(local_out, bounds, rank, dst=0, group=None):
    """Only rank `dst` receives the full [B][width] result (returns None elsewhere): 1/world of the bytes an
    all_gather moves, point-to-point to the root over xGMI — the result gather SURVEY §8e asks for.  Uneven
    shards are padded to the longest one."""
    import torch
    import torch.distributed as dist

    world = len(bounds)
    width = local_out.shape[1]
    longest = max(e - s for s, e in bounds)
    padded = torch.zeros((longest, width), dtype=local_out.dtype, device=local_out.device)
    padded[: local_out.shape[0]] = local_out
    parts = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
    dist.gather(padded, parts, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([parts[r][: e - s] for r, (s, e) in enumerate(bounds)], dim=0)
