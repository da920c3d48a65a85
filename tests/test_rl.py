"""TimeLimit for the vectorised handle and the device ring buffer (rl_utils/replay_buffer.py:27-82)."""
import numpy as np
import pytest
import torch

from hsr_env_amd import GoalSpec, VecHSREnv
from hsr_env_amd.rl import DeviceReplayBuffer, TimeLimit
from oracle_batch import OracleBatchSim


class ListRing:
    """The reference buffer's contract restated with python lists: storage index = (key + pos) % maxlen."""
    def __init__(self, maxlen): self.maxlen, self.store, self.pos, self.full = maxlen, [None] * maxlen, 0, False
    def append_batch(self, items):
        for k, it in enumerate(items): self.store[(self.pos + k) % self.maxlen] = it
        if self.pos + len(items) >= self.maxlen: self.full = True
        self.pos = (self.pos + len(items)) % self.maxlen
    def __len__(self): return self.maxlen if self.full else self.pos
    def get(self, key): return self.store[(key + self.pos) % self.maxlen]


def test_ring_contract_matches_reference_semantics():
    buf, ref = DeviceReplayBuffer(7), ListRing(7)
    assert buf.empty and len(buf) == 0
    rng = np.random.default_rng(0)
    t = 0
    for rounds in range(9):
        b = int(rng.integers(1, 4))
        obs = np.stack([np.full(3, t + k, np.float32) for k in range(b)]); rew = np.arange(t, t + b, dtype=np.float32)
        buf.extend((torch.from_numpy(obs), torch.from_numpy(rew)))
        ref.append_batch([(obs[k], rew[k]) for k in range(b)])
        t += b
        assert len(buf) == len(ref) and buf.pos == ref.pos and buf.full == ref.full
        for key in range(-len(ref), 0):
            o, r = buf[key]
            assert np.array_equal(o.numpy(), ref.get(key)[0]) and float(r) == ref.get(key)[1]
    arr_o, arr_r = buf.array()
    assert np.array_equal(arr_r.numpy(), np.arange(t - 7, t, dtype=np.float32))        # oldest -> newest
    o, r = buf.sample(32)
    assert o.shape == (32, 3) and r.shape == (32,) and r.min() >= t - 7 and r.max() <= t - 1
    o, r = buf.sample(5, seq_len=3)
    assert o.shape == (5, 3, 3) and r.shape == (5, 3)
    single = DeviceReplayBuffer(4)
    for k in range(6):
        single.append({"s": np.full(2, k, np.float32), "a": np.float32(k)})           # one item per call (get_index == 1)
    assert len(single) == 4 and float(single[-1]["a"]) == 5 and float(single[-4]["a"]) == 2 and single[-1]["s"].shape == (2,)


def test_time_limit_truncates_per_env(models):
    m = models["cfg2"]
    env = VecHSREnv(model=m, n_envs=3, sim=OracleBatchSim(m, 3), goals=[GoalSpec("block0", np.array([.4, 0, .422]), .05)], steps_per_action=2)
    env = TimeLimit(env, max_episode_steps=3)                     # hsr/__init__.py:20 uses 20
    env.reset()
    for k in range(3):
        obs, rew, done, info = env.step(np.zeros((3, 2)))
        assert done.all() == (k == 2) and (info["TimeLimit.truncated"] == done).all()
    env.reset(mask=np.array([True, False, False]))
    obs, rew, done, info = env.step(np.zeros((3, 2)))
    assert done.tolist() == [False, True, True]
    assert env.action_space.shape == (2,)                          # attribute passthrough
