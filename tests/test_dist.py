"""Multi-process path on CPU: world_size 2, gloo.  Each rank owns a contiguous env shard (no collective in
the substep loop) and all-gathers the packed obs / reward / done once per env-step."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _worker(rank, world, port, n_global, out_dir):
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hsr_env_amd import GoalSpec, VecHSREnv, Box, load_config
    from hsr_env_amd import dist as hd
    from oracle_batch import OracleBatchSim
    m = load_config("cfg2")
    lo, hi = hd.shard_range(n_global, rank, world)
    env = VecHSREnv(model=m, n_envs=hi - lo, sim=OracleBatchSim(m, hi - lo), env_offset=lo, n_global=n_global, steps_per_action=12,
                    goals=[GoalSpec("block0", Box(low=[-.1, -.2, .422], high=[.1, .2, .422]), .05)],
                    block_space=Box(low=[-.1, -.2, .422, -3], high=[.1, .2, .422, 3]))
    env.seed(5); env.reset()
    rng = np.random.Generator(np.random.Philox(key=[9, 0]))
    act = rng.uniform(-1, 1, (n_global, 2)).astype(np.float32)[lo:hi]
    obs, rew, done, info = env.step(act)
    packed = hd.pack_step(torch.from_numpy(np.atleast_2d(obs)), torch.from_numpy(np.atleast_1d(rew)), torch.from_numpy(np.atleast_1d(done)))
    allp = hd.all_gather_step(packed, world)
    o, r, d = hd.unpack_step(allp)
    if rank == 0:
        np.savez(Path(out_dir) / "gathered.npz", obs=o.numpy(), rew=r.numpy(), done=d.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_global", [4, 5])       # equal and ragged shards
def test_sharded_step_all_gather_matches_single_process(tmp_path, n_global):
    port = 29500 + os.getpid() % 500 + n_global
    mp.spawn(_worker, args=(2, port, n_global, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "gathered.npz")
    sys.path.insert(0, str(ROOT / "tests"))
    from hsr_env_amd import GoalSpec, VecHSREnv, Box, load_config
    from oracle_batch import OracleBatchSim
    m = load_config("cfg2")
    env = VecHSREnv(model=m, n_envs=n_global, sim=OracleBatchSim(m, n_global), steps_per_action=12,
                    goals=[GoalSpec("block0", Box(low=[-.1, -.2, .422], high=[.1, .2, .422]), .05)],
                    block_space=Box(low=[-.1, -.2, .422, -3], high=[.1, .2, .422, 3]))
    env.seed(5); env.reset()
    rng = np.random.Generator(np.random.Philox(key=[9, 0]))
    act = rng.uniform(-1, 1, (n_global, 2)).astype(np.float32)
    obs, rew, done, info = env.step(act)
    assert np.array_equal(got["obs"], obs) and np.array_equal(got["rew"], rew) and np.array_equal(got["done"], done)


def test_shard_range_partitions():
    from hsr_env_amd.dist import shard_range
    for n in (1, 7, 8192, 65536):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` as a plain command (no torch.distributed.run): the parent starts N ranks with an env://
    rendezvous on 127.0.0.1, relays rank 0's line and returns the worst exit code - checked here without a GPU."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "3", "--launch-check"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["RANK"] == "0" and line["WORLD_SIZE"] == "3" and line["MASTER_ADDR"] == "127.0.0.1" and int(line["MASTER_PORT"]) > 0
    # a failing rank fails the launch AND takes the others down: rank 1 exits at once, rank 0 and 2 would wait two minutes (like a
    # rank stuck in the rendezvous of a peer that died) - the launcher returns the failure within seconds and leaves no child behind
    import time
    t0 = time.time()
    r = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "3", "--launch-check", "--fail-rank", "1"], env=env, capture_output=True, text=True, timeout=100)
    assert r.returncode == 3, (r.returncode, r.stderr)
    assert time.time() - t0 < 30
