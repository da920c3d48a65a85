"""Consumers of the batched env outputs: the `TimeLimit(env, 20)` wrapper of hsr/__init__.py:20 for
the vectorised handle, and the ring `ReplayBuffer` of rl_utils/replay_buffer.py:27-82 kept on the device, so that transitions
produced by `BatchSim.step_dev` never visit the host.  Plumbing around the hot path: torch tensors, no kernels."""
from __future__ import annotations

from typing import Any

import numpy as np


class TimeLimit:
    """gym.wrappers.TimeLimit for N envs: an env whose episode reaches `max_episode_steps` env-steps reports done with
    info['TimeLimit.truncated'] = not already-done (per env); the counter of an env restarts when it is reset."""

    def __init__(self, env, max_episode_steps: int):
        self.env = env
        self._max_episode_steps = int(max_episode_steps)
        self._elapsed = np.zeros(env.n_envs, dtype=np.int64)

    def __getattr__(self, name):
        return getattr(self.env, name)

    def reset(self, *args, **kwargs):
        mask = kwargs.get("mask", args[0] if args else None)
        if mask is None:
            self._elapsed[:] = 0
        else:
            self._elapsed[np.asarray(mask, bool)] = 0
        return self.env.reset(*args, **kwargs)

    def step(self, action, *args, **kwargs):
        obs, rew, done, info = self.env.step(action, *args, **kwargs)
        self._elapsed += 1
        over = self._elapsed >= self._max_episode_steps
        d = np.atleast_1d(np.asarray(done, bool))
        info = dict(info)
        trunc = over & ~d
        if self.env.n_envs == 1:
            info["TimeLimit.truncated"] = bool(trunc[0])
            return obs, rew, bool(d[0] or over[0]), info
        info["TimeLimit.truncated"] = trunc
        return obs, rew, d | over, info


def _leaves(x):
    if isinstance(x, dict):
        return list(x.values())
    if isinstance(x, (tuple, list)):
        return list(x)
    return [x]


def _like(x, leaves):
    if isinstance(x, dict):
        return dict(zip(x.keys(), leaves))
    if isinstance(x, tuple) and hasattr(x, "_fields"):
        return type(x)(*leaves)
    if isinstance(x, (tuple, list)):
        return type(x)(leaves)
    return leaves[0]


class DeviceReplayBuffer:
    """rl_utils/replay_buffer.py:27-82 with torch tensors on `device`.  Same indexing contract: keys are RELATIVE to the
    write position `pos` (-1 = newest item, -len = oldest), storage index = (key + pos) % maxlen; `append` takes one item or
    a batch (leading dimension shared by all leaves, `get_index` of the reference) and advances `pos` by that count."""

    def __init__(self, maxlen: int, device: Any = "cpu", seed: int = 0):
        import torch
        self.maxlen, self.device = int(maxlen), torch.device(device)
        self.buffer = None
        self.full = False
        self.pos = 0
        self._proto = None
        self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed(seed)

    @property
    def empty(self):
        return self.buffer is None

    def __len__(self):
        return self.maxlen if self.full else self.pos

    def _t(self, v):
        import torch
        return v.to(self.device) if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v), device=self.device)

    def modulate(self, key):
        import torch
        if isinstance(key, slice):
            key = torch.arange(key.start or 0, 0 if key.stop is None else key.stop, key.step or 1, device=self.device)
        return (self._t(key) + self.pos) % self.maxlen

    def __getitem__(self, key):
        assert self.buffer is not None
        idx = self.modulate(key)
        return _like(self._proto, [b[idx] for b in self.buffer])

    def array(self):
        return self[-len(self):0] if self.buffer is not None else None

    def _count(self, leaves):
        sizes = {int(v.shape[0]) if v.dim() > 0 else 1 for v in leaves}
        return sizes.pop() if len(sizes) == 1 else 1

    def append(self, x, batched: bool = None):
        """One item (leaves without batch dimension) or a batch of items (all leaves share the leading dimension)."""
        import torch
        leaves = [self._t(v) for v in _leaves(x)]
        stop = self._count(leaves) if batched is None else (int(leaves[0].shape[0]) if batched else 1)
        is_batch = batched if batched is not None else (stop > 1 or all(v.dim() > 0 and v.shape[0] == 1 for v in leaves) and self.buffer is not None
                                                         and all(v.dim() == b.dim() for v, b in zip(leaves, self.buffer)))
        if self.buffer is None:
            self._proto = x
            self.buffer = [torch.zeros((self.maxlen,) + tuple(v.shape[1:] if is_batch else v.shape), dtype=v.dtype, device=self.device) for v in leaves]
        if stop > self.maxlen:
            raise ValueError("batch larger than the buffer")
        idx = (torch.arange(stop, device=self.device) + self.pos) % self.maxlen
        for b, v in zip(self.buffer, leaves):
            b[idx] = v if is_batch else v.unsqueeze(0)
        if self.pos + stop >= self.maxlen:
            self.full = True
        self.pos = (self.pos + stop) % self.maxlen

    def extend(self, x):
        self.append(x, batched=True)

    def sample(self, batch_size: int, seq_len: int = None):
        import torch
        n = len(self)
        assert n > 0
        idx = torch.randint(-n, 0, (batch_size,), device=self.device, generator=self._gen)
        if seq_len is not None:
            idx = idx[:, None] + torch.arange(seq_len, device=self.device)[None, :]
        return self[idx]
