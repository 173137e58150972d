"""Batch sharding over the GPUs of one node.

Paths are independent, so the batch is cut into contiguous blocks of path
indices, one block per rank (one process per GPU). Inputs are generated /
uploaded per shard; the only collective is ONE gather of the packed timing
profile (time, s, sd, sdd: [4][B_shard][N] fp64) to rank 0 at the end, issued on
torch.distributed's default backend (RCCL over xGMI when the backend is
"nccl"; "gloo" in the CPU tests). Ragged shards use one grouped send/recv.
"""
import torch
import torch.distributed as dist


def shard_bounds(total, world_size, rank):
    """Contiguous [lo, hi) block of `total` items for `rank`; sizes differ by at most 1."""
    base, rem = divmod(int(total), int(world_size))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def balanced_bounds(costs, world_size):
    """Contiguous partition of items with per-item `costs` (e.g. N*C^2 for ragged
    batches) into world_size blocks of roughly equal total cost.
    Returns a list of (lo, hi)."""
    total = float(sum(costs))
    bounds, lo, acc, k = [], 0, 0.0, 1
    n = len(costs)
    for i, c in enumerate(costs):
        acc += float(c)
        remaining_items = n - (i + 1)
        remaining_blocks = world_size - k
        if k < world_size and (acc >= total * k / world_size or remaining_items == remaining_blocks):
            bounds.append((lo, i + 1))
            lo = i + 1
            k += 1
    bounds.append((lo, n))
    while len(bounds) < world_size:
        bounds.append((n, n))
    return bounds[:world_size]


def gather_packed(packed, dst=0, group=None):
    """ONE gather of equally-shaped shard tensors to `dst`.

    packed: contiguous tensor (same shape on every rank). Returns on dst a tensor
    of shape [world_size, *packed.shape]; None elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return packed.unsqueeze(0)
    if rank == dst:
        full = torch.empty((world,) + tuple(packed.shape), dtype=packed.dtype,
                           device=packed.device)
        dist.gather(packed, gather_list=list(full.unbind(0)), dst=dst, group=group)
        return full
    dist.gather(packed, gather_list=None, dst=dst, group=group)
    return None


class PipelinedGather:
    """The single gather per batch, overlapped with the next batch's solve.

    Rank `dst` receives 7/8 of every batch through its inbound xGMI links (64 KB per path
    for the packed timing profile), which takes about as long as solving a shard
    (SURVEY.md 8e), so the gather of batch k runs asynchronously while batch k+1 is being
    solved into the other of `depth` output buffers. Usage per batch k:
        buf = g.buffer(k)          # waits until the gather that last read it is done
        ... solve into buf ...
        g.launch(k)                # ONE async gather of buf (RCCL stream)
    and g.drain() before results are read / timing stops. On `dst`, result(k) is the
    [world, *shape] tensor of batch k (valid after the gather completed)."""

    def __init__(self, shape, dtype, device, depth=2, dst=0, group=None, on_complete=None):
        """on_complete(slot): called on `dst` right after the gather into recv[slot] has been
        waited for (the current stream is then ordered behind it) -- e.g. to launch what the
        root derives from the payload (bench.py: the time samples rebuilt from sd)."""
        self.group, self.dst, self.depth = group, dst, depth
        self.on_complete = on_complete
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.send = [torch.empty(shape, dtype=dtype, device=device) for _ in range(depth)]
        self.recv = None
        if self.world > 1 and self.rank == dst:
            self.recv = [torch.empty((self.world,) + tuple(shape), dtype=dtype, device=device)
                         for _ in range(depth)]
        self.work = [None] * depth
        self.batch = [0] * depth            # batch number of the gather in flight per slot

    def buffer(self, k, host_sync=False):
        """host_sync: also block the HOST until the gather that last read this buffer is done.
        Needed when the next writer is not ordered on the current stream -- the engine's pipelined
        mode writes q from its own stream (include/tpamd.h tpamd_engine_set_pipelining)."""
        slot = k % self.depth
        if self.work[slot] is not None:
            self.work[slot].wait()      # stream-level wait for NCCL, blocking wait for gloo
            self.work[slot] = None
            if self.on_complete is not None and self.rank == self.dst:
                self.on_complete(slot)
            if host_sync and self.send[slot].is_cuda:
                torch.cuda.current_stream(self.send[slot].device).synchronize()
        return self.send[slot]

    def launch(self, k):
        slot = k % self.depth
        self.batch[slot] = k
        if self.world == 1:
            return
        if self.rank == self.dst:
            self.work[slot] = dist.gather(self.send[slot], gather_list=list(self.recv[slot].unbind(0)),
                                          dst=self.dst, group=self.group, async_op=True)
        else:
            self.work[slot] = dist.gather(self.send[slot], gather_list=None, dst=self.dst,
                                          group=self.group, async_op=True)

    def drain(self):
        for slot in sorted(range(self.depth), key=lambda sl: self.batch[sl]):   # oldest batch first
            if self.work[slot] is not None:
                self.work[slot].wait()
                self.work[slot] = None
                if self.on_complete is not None and self.rank == self.dst:
                    self.on_complete(slot)

    def result(self, k):
        slot = k % self.depth
        if self.world == 1:
            return self.send[slot].unsqueeze(0)
        return self.recv[slot] if self.rank == self.dst else None


def gather_ragged(shard, counts, dst=0, group=None):
    """Gather shards whose leading dimension differs per rank (counts[r] rows on
    rank r, known to every rank) with one grouped send/recv. Returns the
    concatenation on dst, None elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return shard
    if rank == dst:
        total = int(sum(counts))
        full = torch.empty((total,) + tuple(shard.shape[1:]), dtype=shard.dtype,
                           device=shard.device)
        offs = [0]
        for c in counts:
            offs.append(offs[-1] + int(c))
        full[offs[dst]:offs[dst + 1]].copy_(shard)
        ops = [dist.P2POp(dist.irecv, full[offs[r]:offs[r + 1]], r, group)
               for r in range(world) if r != dst and counts[r] > 0]
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return full
    if shard.shape[0] > 0:
        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, shard.contiguous(), dst, group)]):
            w.wait()
    return None
