"""Env sharding across the GPUs of one node (one process per GPU, torch.distributed; backend "nccl"
is RCCL on ROCm, "gloo" in CPU tests).

Envs are independent (one MjSim per env in the reference, hsr/control.py:67), so the substep loop needs
no collective: rank g owns the contiguous global env range [g*N/G, (g+1)*N/G).  The only exchange is one
all-gather per env-step of the returned obs / reward / done, packed into a single
fp32 buffer [N_local, nq+nv+2] so that it is one collective of ~1.8 MB per GPU at 8192 envs.
"""
from __future__ import annotations

import os
from typing import Tuple


def rank_world() -> Tuple[int, int, int]:
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def shard_range(n_global: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block partition; the first n_global % world ranks hold one extra env."""
    base, rem = divmod(n_global, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_step(obs, reward, done):
    """[N_local, nq+nv] + [N_local] + [N_local] -> one fp32 tensor [N_local, nq+nv+2]."""
    import torch
    return torch.cat([obs, reward.reshape(-1, 1).to(obs.dtype), done.reshape(-1, 1).to(obs.dtype)], dim=1).contiguous()


def unpack_step(packed):
    return packed[:, :-2], packed[:, -2], packed[:, -1] > 0.5


def all_gather_step(packed_local, world: int, out=None, equal_shards=None):
    """One all-gather of the packed step outputs.  Equal shard sizes (the weak-scaling benchmark) use the flat
    all_gather_into_tensor; ragged shards exchange sizes first and pad (pass equal_shards to skip the size exchange)."""
    import torch
    import torch.distributed as dist
    if world == 1 or not dist.is_initialized():
        return packed_local
    if equal_shards is None:
        n_local = torch.tensor([packed_local.shape[0]], device=packed_local.device)
        sizes = [torch.zeros_like(n_local) for _ in range(world)]
        dist.all_gather(sizes, n_local)
        sizes = [int(s.item()) for s in sizes]
        equal_shards = len(set(sizes)) == 1
    else:
        sizes = [packed_local.shape[0]] * world
    if equal_shards:
        if out is None:
            out = torch.empty((world * packed_local.shape[0], packed_local.shape[1]), dtype=packed_local.dtype, device=packed_local.device)
        if dist.get_backend() == "gloo":
            bufs = list(out.chunk(world, dim=0))
            dist.all_gather(bufs, packed_local)
        else:
            dist.all_gather_into_tensor(out, packed_local)
        return out
    mx = max(sizes)
    pad = torch.zeros((mx, packed_local.shape[1]), dtype=packed_local.dtype, device=packed_local.device)
    pad[:packed_local.shape[0]] = packed_local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return torch.cat([b[:n] for b, n in zip(bufs, sizes)], dim=0)


class StepGather:
    """The per-env-step exchange of the sharded run: pack obs / reward / done of the local shard into one fp32
    buffer [N_local, nq+nv+2] and all-gather it into [world * N_local, nq+nv+2].  The packing runs on torch's CURRENT stream - bench.py
    makes that the batch's own HIP stream (torch.cuda.ExternalStream), so it is ordered after the env-step kernel without a host sync.
    Equal shards only (the weak-scaling layout); the buffers are allocated once.  With `always=True` the collective is issued even
    for a single rank (RCCL accepts a one-rank communicator): the GPU test uses that to execute this exact code path on one GPU.

    overlap=False: the collective is issued on the current stream too - the next env-step of this rank starts when every rank has delivered
    this one (a rank-wide barrier per env-step: the sharded run then advances at the pace of the slowest rank of EVERY step).
    overlap=True (RCCL only): the collective runs on a side stream, behind an event recorded after the packing, into one of two buffer pairs;
    the current stream goes on with the next env-step and only waits - two steps later, before it packs into the same pair again - for the
    collective that read it.  Ranks drift by up to two env-steps instead of meeting at every one; the gathered buffer of a step is valid
    once `wait(result)` has been called (a stream wait, no host sync) or the device has been synchronised."""

    def __init__(self, n_local: int, nobs: int, world: int, device, always: bool = False, overlap: bool = False):
        import torch
        self.world, self.nobs, self.always = world, nobs, always
        exchange = world > 1 or always
        self.overlap = bool(overlap and exchange)
        nbuf = 2 if self.overlap else 1
        self.packs = [torch.empty((n_local, nobs + 2), dtype=torch.float32, device=device) for _ in range(nbuf)]
        self.alls = [torch.empty((world * n_local, nobs + 2), dtype=torch.float32, device=device) if exchange else self.packs[i] for i in range(nbuf)]
        self.pack, self.all = self.packs[0], self.alls[0]
        self.side = torch.cuda.Stream(device=device) if self.overlap else None
        self.done_ev = [None] * nbuf          # fired when the collective that used buffer pair i has finished
        self.k = 0

    def wait(self, gathered=None):
        """Make the current stream wait for the collective that fills `gathered` (default: every collective issued so far)."""
        if not self.overlap:
            return
        import torch
        cur = torch.cuda.current_stream()
        for i, ev in enumerate(self.done_ev):
            if ev is not None and (gathered is None or gathered.data_ptr() == self.alls[i].data_ptr()):
                cur.wait_event(ev)

    def __call__(self, obs, reward, done):
        import torch
        import torch.distributed as dist
        nobs = self.nobs
        i = self.k % len(self.packs)
        self.k += 1
        if self.overlap and self.done_ev[i] is not None:
            torch.cuda.current_stream().wait_event(self.done_ev[i])          # the collective of two steps ago has read pack[i] / written all[i]
        pack, out = self.packs[i], self.alls[i]
        pack[:, :nobs] = obs
        pack[:, nobs] = reward
        pack[:, nobs + 1] = done
        if (self.world > 1 or self.always) and dist.is_initialized():
            if dist.get_backend() == "gloo":
                host = pack.cpu()
                bufs = [host.new_empty(host.shape) for _ in range(self.world)]
                dist.all_gather(bufs, host)
                out.copy_(torch.cat(bufs, dim=0))
            elif self.overlap:
                packed = torch.cuda.Event()
                packed.record(torch.cuda.current_stream())
                with torch.cuda.stream(self.side):
                    self.side.wait_event(packed)
                    dist.all_gather_into_tensor(out, pack)                   # torch runs it on its RCCL stream, ordered behind and before `side`
                    ev = torch.cuda.Event()
                    ev.record(self.side)
                self.done_ev[i] = ev
            else:
                dist.all_gather_into_tensor(out, pack)
        return out


def init_process_group(backend: str, device=None):
    """env:// rendezvous (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the launcher); "nccl" is RCCL on ROCm and is bound to
    `device` at creation (device_id), as the eager-init path of torch requires for collectives on a non-default stream."""
    import torch.distributed as dist
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=device)
    else:
        dist.init_process_group(backend)
