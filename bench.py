#!/usr/bin/env python
"""Headline benchmark: env-steps/s of the batched HSR stepper (BASELINE.json metric).

One "step" = one env-step of every env on every rank: ctrl write, 300 substeps of h = 0.002 with the
per-substep goal test / early exit (hsr/env.py:115-135), obs/reward/done, the `if done: reset()` of the
reference's driver loop (hsr/control.py:73-75) and, for N > 1, one RCCL all-gather of obs/reward/done.
Workload (config.workload): BASELINE config 3 - all 7 DOFs + 1 block, 8192 envs per GPU (weak scaling),
synthetic inputs of SURVEY.md section 8d resident in HBM before the timed region.

    python bench.py [--gpus N --steps K --warmup W]
N > 1 works both ways: under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (RANK / WORLD_SIZE in
the environment) and as the plain command above, which then starts the N ranks itself (one child process per GPU; the parent
never touches the GPU) and relays rank 0's JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

STEPS_PER_ACTION = 300
GEOFENCE = 0.05
HBM_PEAK_GBPS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def sample_inputs(m, n, seed, offset):
    """Synthetic inputs of SURVEY.md 8d for global env ids [offset, offset+n): block start
    (x,y) ~ U([-.1,.1]x[-.2,.2]), z = .422, yaw ~ U(-pi,pi); goal ~ U(same box) but at least
    2*geofence from the block so that an early exit needs the robot to push the block there."""
    rng = np.random.Generator(np.random.Philox(key=[seed, offset]))
    q = np.tile(m.qpos0, (n, 1))
    blocks = m.free_joint_qadrs()
    nb = len(blocks)
    for b, a in enumerate(blocks):
        yaw = rng.uniform(-np.pi, np.pi, n)
        q[:, a] = rng.uniform(-0.1, 0.1, n)
        q[:, a + 1] = rng.uniform(-0.2, 0.2, n) if nb == 1 else rng.uniform(-0.04, 0.04, n) + 0.13 * (b - (nb - 1) / 2)
        q[:, a + 2] = 0.422
        q[:, a + 3] = np.cos(yaw / 2); q[:, a + 4] = 0; q[:, a + 5] = 0; q[:, a + 6] = np.sin(yaw / 2)
    goal = np.column_stack([rng.uniform(-0.1, 0.1, n), rng.uniform(-0.2, 0.2, n), np.full(n, 0.422)])
    if nb:
        a0 = blocks[0]
        for _ in range(50):
            close = np.linalg.norm(goal[:, :2] - q[:, a0:a0 + 2], axis=1) < 2 * GEOFENCE
            if not close.any():
                break
            goal[close, 0] = rng.uniform(-0.1, 0.1, close.sum()); goal[close, 1] = rng.uniform(-0.2, 0.2, close.sum())
    return q.astype(np.float32), goal.astype(np.float32)


def host_cores():
    """Threads this process may really use: the affinity mask capped by the cgroup CPU quota."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = max(1, min(cores, int(np.ceil(int(quota) / int(period)))))
    except (OSError, ValueError):
        pass
    return cores


def cpu_baseline(m, q0, goal, ctrl, cores):
    """Oracle (fp64 C restatement, OpenMP over envs) on the host cores, bounded sample of the same workload:
    a calibration pass on 8 envs per thread sizes the sample (first n envs, r env-steps) to about 12 s; then the same
    code on ONE thread for about 5 s (single-core and all-core)."""
    from oracle import oracle as orc
    bid = m.body_id(m.block_body()) if m.block_body() else -1

    def run(n, reps, threads=cores):
        qpos = q0[:n].astype(np.float64).copy(); qvel = np.zeros((n, m.nv)); warm = np.zeros((n, m.nv))
        mocap = goal[:n].astype(np.float64).copy()
        t0 = time.perf_counter()
        for k in range(reps):
            orc.batch_env_step(m, qpos, qvel, warm, ctrl[k % len(ctrl)][:n].astype(np.float64).copy(), mocap, STEPS_PER_ACTION, bid, GEOFENCE, threads)
        return time.perf_counter() - t0

    N = q0.shape[0]
    n_cal = min(N, 8 * cores)
    per_env_step = run(n_cal, 1) / n_cal                  # seconds of wall time per env-step at this thread count
    budget = 12.0
    n = int(min(N, max(n_cal, budget / max(per_env_step, 1e-9))))
    reps = int(max(1, min(8, budget / max(per_env_step * n, 1e-9))))
    dt = run(n, reps)
    t1 = run(min(N, 16), 1, threads=1) / min(N, 16)                       # calibrate one thread, then about 5 s on it
    n1 = int(max(1, min(N, 5.0 / max(t1, 1e-9))))
    dt1 = run(n1, 1, threads=1)
    eff = (n * reps / dt) / (cores * (n1 / dt1))
    all_threads = {"value": n * reps / dt, "unit": "env-steps/s", "cores": cores, "parallel_efficiency": eff,
                   "sample": f"first {n} envs x {reps} env-steps x {STEPS_PER_ACTION} substeps of the same workload ({dt:.1f} s), OpenMP over envs ({cores} threads)"}
    single = {"value": n1 / dt1, "unit": "env-steps/s", "cores": 1, "sample": f"first {n1} envs x 1 env-step x {STEPS_PER_ACTION} substeps of the same workload ({dt1:.1f} s)"}
    what = "oracle/hsr_oracle.c (fp64 restatement); stand-in for CPU mujoco-py, which is not installable here"
    if eff < 0.5:
        # a box that does not deliver its nominal threads to this process (shared or throttled host: 1.6x one thread on "16 cores" in rounds 4-5):
        # the headline of the baseline is then the ONE-core measurement, `cores` says 1, and the all-thread run is kept beside it
        all_threads["note"] = (f"the {cores} threads deliver {eff * cores:.1f}x one thread: this process does not get {cores} cores' worth of CPU on this box; "
                               "not an all-core figure - `value` above is the single-core measurement")
        return dict(value=single["value"], unit="env-steps/s", cores=1, kind="port", sample=single["sample"] + ", " + what, all_threads=all_threads)
    return dict(value=all_threads["value"], unit="env-steps/s", cores=cores, kind="port", parallel_efficiency=eff, single_core=single,
                sample=all_threads["sample"] + ", " + what)


def capacity_leg(m, n, dev, device_id, K=3, W=2):
    """SECONDARY figure, never the headline: the same env-step at `n` envs on this one GPU (tasks beyond the resident workgroups go through the work
    queue), K timed env-steps after W warm-up ones - what the GPU delivers once the serial chain of the hardest env is amortised over more envs."""
    import torch
    from hsr_env_amd.sim import BatchSim
    q0, goal = sample_inputs(m, n, 0, 0)
    rng = np.random.Generator(np.random.Philox(key=[1, 0]))
    lo, hi = m.act_ctrlrange[:, 0].astype(np.float32), m.act_ctrlrange[:, 1].astype(np.float32)
    sim = BatchSim(m, n, device=device_id)
    sim.reset(qpos0=q0, mocap=goal)
    ext = torch.cuda.ExternalStream(sim.stream_ptr(), device=dev)
    bid = m.body_id(m.block_body()) if m.block_body() else -1
    with torch.cuda.stream(ext):
        d_ctrl = [torch.from_numpy(rng.uniform(lo, hi, (n, m.nu)).astype(np.float32)).to(dev) for _ in range(K + W)]
        rs = [sample_inputs(m, n, 2 + k, 0) for k in range(K + W)]
        d_rq = [torch.from_numpy(r[0]).to(dev) for r in rs]; d_rg = [torch.from_numpy(r[1]).to(dev) for r in rs]
        d_obs = torch.empty((n, m.nq + m.nv), dtype=torch.float32, device=dev); d_rew = torch.empty(n, dtype=torch.float32, device=dev)
        d_done = torch.empty(n, dtype=torch.uint8, device=dev); d_ns = torch.empty(n, dtype=torch.int32, device=dev)

        def env_step(k):
            sim.step_dev(d_ctrl[k].data_ptr(), STEPS_PER_ACTION, bid, GEOFENCE, d_obs.data_ptr(), d_rew.data_ptr(), d_done.data_ptr(), d_ns.data_ptr())
            sim.reset_dev(None, d_rq[k].data_ptr(), d_rg[k].data_ptr())
        for k in range(W):
            env_step(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(W, W + K):
            env_step(k)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    bad, _ = sim.bad_state()
    sim.close()
    return {"envs_per_gpu": n, "value": n * K / dt, "unit": "env-steps/s", "ms_per_step": 1e3 * dt / K, "steps": K, "warmup": W, "bad_envs": int(bad.sum()),
            "note": "secondary: throughput of ONE GPU at saturation (same workload, more envs per GPU, work queue); not the BASELINE configuration, never the headline"}


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (env:// rendezvous on 127.0.0.1), relay rank
    0's output, fail if any rank fails.  Runs before torch / HIP are touched - the parent process never initialises the GPU.
    The ranks are polled together: the first one that fails takes the others down with it (a rank whose peer died would sit in the
    rendezvous until its timeout, holding its GPU), and none survives the parent."""
    import socket
    import subprocess
    import tempfile
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    out0 = tempfile.TemporaryFile(mode="w+")
    try:
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=out0 if r == 0 else subprocess.DEVNULL, text=True))
        worst = 0
        live = list(procs)
        while live:
            for p in list(live):
                rc = p.poll()
                if rc is None:
                    continue
                live.remove(p)
                if rc != 0 and worst == 0:
                    worst = rc
                    for q in live:
                        q.terminate()
            if live:
                time.sleep(0.05)
        out0.seek(0)
        sys.stdout.write(out0.read())
        sys.stdout.flush()
        return worst
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            try:
                p.wait(timeout=10)
            except Exception:
                pass
        out0.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--envs-per-gpu", type=int, default=8192)
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--gather-overlap", action="store_true", help="N > 1: issue the all-gather of env-step k on a side stream, overlapped with env-step k + 1 (ranks may drift "
                    "by two env-steps; only legal when step k + 1 does not depend on the gathered obs of step k - an OPEN-loop number).  Default: the all-gather in "
                    "line on the batch stream, every rank waits for it before its next env-step (closed loop: what a trainer that acts on the gathered obs needs)")
    ap.add_argument("--no-gather-overlap", action="store_true", help=argparse.SUPPRESS)      # accepted for the command lines of round 4: in-line is the default now
    ap.add_argument("--no-capacity", action="store_true", help="skip the secondary `capacity` leg (N = 1: a few env-steps at --capacity-envs envs on this GPU after the headline)")
    ap.add_argument("--capacity-envs", type=int, default=65536)
    ap.add_argument("--no-persistent", action="store_true", help="per-substep kernels instead of the persistent env-step kernel")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="control-flow rehearsal of the N>1 path on a box with one GPU: every rank uses cuda:0 and the all-gather "
                         "goes through gloo on host copies (RCCL refuses two ranks on one device); the number it prints is not a result")
    ap.add_argument("--env-offset", type=int, default=0, help="one-rank runs: global id of this process's first env (the shard a rank of a "
                    "multi-GPU run would own); inputs are keyed by global env id, so the shard is reproduced env for env")
    ap.add_argument("--dump-step", default=None, help="rank 0 writes obs | reward | done of the LAST timed env-step to this .npy: the gathered "
                    "[global_envs, nq+nv+2] buffer for N > 1, the local one otherwise (tests compare the two)")
    ap.add_argument("--launch-check", action="store_true", help="print this rank's rendezvous environment and exit (no GPU touched): "
                    "exercises the self-launcher of --gpus N on a machine without GPUs")
    ap.add_argument("--fail-rank", type=int, default=-1, help="with --launch-check: this rank exits with code 3 at once and the others "
                    "wait (as a rank does whose peer never reaches the rendezvous) - the launcher has to take them down")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    if args.launch_check:
        if args.fail_rank >= 0:
            if int(os.environ.get("RANK", 0)) == args.fail_rank:
                raise SystemExit(3)
            time.sleep(120)
        print(json.dumps({k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}))
        return

    import torch
    from hsr_env_amd.compiler import load_config
    from hsr_env_amd.sim import BatchSim
    from hsr_env_amd import dist as hdist

    rank, world, local_rank = hdist.rank_world()
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        hdist.init_process_group("gloo" if args.rehearse_on_one_gpu else "nccl", dev)

    m = load_config(args.config)
    n = args.envs_per_gpu
    offset = rank * n + (args.env_offset if world == 1 else 0)
    K, W = args.steps, args.warmup
    total = K + W
    q0, goal = sample_inputs(m, n, 0, offset)
    rng = np.random.Generator(np.random.Philox(key=[1, offset]))
    lo, hi = m.act_ctrlrange[:, 0].astype(np.float32), m.act_ctrlrange[:, 1].astype(np.float32)
    ctrl_host = [rng.uniform(lo, hi, (n, m.nu)).astype(np.float32) for _ in range(total)]
    reset_host = [sample_inputs(m, n, 2 + k, offset) for k in range(total)]

    sim = BatchSim(m, n, device=local_rank)
    if args.no_graph:
        sim.set_graph(False)
    if args.no_persistent:
        sim.set_persistent(False)
    persistent = sim.is_persistent()
    sim.reset(qpos0=q0, mocap=goal)
    # everything the timed region consumes is resident in HBM before it starts
    d_ctrl = [torch.from_numpy(c).to(dev) for c in ctrl_host]
    d_rq = [torch.from_numpy(r[0]).to(dev) for r in reset_host]
    d_rg = [torch.from_numpy(r[1]).to(dev) for r in reset_host]
    nobs = m.nq + m.nv
    d_obs = torch.empty((n, nobs), dtype=torch.float32, device=dev)
    d_rew = torch.empty(n, dtype=torch.float32, device=dev)
    # done / nsteps of every env-step are kept (5 bytes per env and step) and summed after the timed region: no reduction kernels in it
    d_done = torch.empty((total, n), dtype=torch.uint8, device=dev)
    d_ns = torch.empty((total, n), dtype=torch.int32, device=dev)
    bid = m.body_id(m.block_body()) if m.block_body() else -1

    # torch ops (packing, the RCCL all-gather) are enqueued on the batch's own stream
    ext = torch.cuda.ExternalStream(sim.stream_ptr(), device=dev)
    torch.cuda.set_stream(ext)
    # closed loop by default: the collective of env-step k is in line on the batch stream; --gather-overlap moves it to a side stream under env-step k + 1
    # (RCCL only; the gloo rehearsal goes through host copies and stays in line)
    gather_overlap = world > 1 and not args.rehearse_on_one_gpu and args.gather_overlap
    gather = hdist.StepGather(n, nobs, world, dev, overlap=gather_overlap) if world > 1 else None

    last_gathered = [None]

    def env_step(k):
        sim.step_dev(d_ctrl[k].data_ptr(), STEPS_PER_ACTION, bid, GEOFENCE, d_obs.data_ptr(), d_rew.data_ptr(),
                     d_done[k].data_ptr(), d_ns[k].data_ptr())
        if gather is not None:
            last_gathered[0] = gather(d_obs, d_rew, d_done[k])          # obs / reward / done of every rank's shard
        sim.reset_dev(None, d_rq[k].data_ptr(), d_rg[k].data_ptr())     # `if done: env.reset()`

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(W):
        env_step(k)
    barrier()
    sim.cap_counts()                        # clears the counters: the cap statistics below cover the timed steps only
    sim.set_profiling(2)                    # an event pair around every launch of the persistent kernel, no synchronisation
    barrier()
    t0 = time.perf_counter()
    for k in range(W, W + K):
        env_step(k)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.rehearse_on_one_gpu else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if args.dump_step and rank == 0:
        if last_gathered[0] is not None:
            np.save(args.dump_step, last_gathered[0].detach().cpu().numpy())
        else:
            np.save(args.dump_step, np.concatenate([d_obs.cpu().numpy(), d_rew.cpu().numpy()[:, None], d_done[W + K - 1].cpu().numpy()[:, None].astype(np.float32)], axis=1))
    substeps_done = int(d_ns[W:W + K].sum().item())
    dones = int(d_done[W:W + K].sum().item())

    # launch durations of the dominant kernel over the timed region (HIP events on the batch stream, one pair per launch)
    if persistent:
        # one launch = one env-step of every env (kinematics + collision + solve + integrate, 300 substeps in-kernel)
        k_all = sim.kernel_times(cap=K)         # also the first synchronising call after the timed region: raises if a launch was drained by the work queue's watchdog
        sim.set_profiling(False)
        assert len(k_all) == K, f"{len(k_all)} launches of the persistent kernel were timed, {K} env-steps ran"
        names = ["-", "-", "k_env_step_mf"]
        dom = 2
        units_per_launch = STEPS_PER_ACTION
        avg_us = 1e3 * float(k_all.mean())
        k_ms = [0.0, 0.0, float(k_all.mean())]; k_n = [0, 0, 1]
        kernel_stats = {"kernel_ms_mean": float(k_all.mean()), "kernel_ms_min": float(k_all.min()), "kernel_ms_max": float(k_all.max()),
                        "launches_timed": int(len(k_all)), "outside_kernel_ms": 1e3 * dt / K - float(k_all.mean())}
    else:
        # per-substep chain: one extra profiled env-step (its event log synchronises, so it stays outside the timed region)
        sim.set_profiling(True)
        sim.step_dev(d_ctrl[W + K - 1].data_ptr(), STEPS_PER_ACTION, bid, GEOFENCE, d_obs.data_ptr(), d_rew.data_ptr(), d_done[0].data_ptr(), d_ns[0].data_ptr())
        sim.sync()
        tot_ms, k_ms, k_n = sim.last_timing()
        sim.set_profiling(False)
        names = ["k_kinematics", "k_cull+k_narrow", "k_solve_mf"]
        dom = int(np.argmax(k_ms))
        units_per_launch = 1
        avg_us = 1e3 * k_ms[dom] / max(k_n[dom], 1)
        kernel_stats = {}
    mean_substeps = substeps_done / (n * K)
    # algorithmic bytes: fp32 state stream per substep per env = 4*(2nq+2nv+nu+3) (SURVEY.md 8d / BASELINE.md 3), times the
    # substeps a launch really ran (early exits shorten it), plus the obs / reward / done write of the persistent launch
    bytes_per_substep_env = 4 * (2 * m.nq + 2 * m.nv + m.nu + 3)
    bytes_per_launch = bytes_per_substep_env * n * (mean_substeps if persistent else 1) + (4 * (m.nq + m.nv) + 5) * n * (1 if persistent else 0)
    achieved = bytes_per_launch / (avg_us * 1e-6) / 1e9
    traffic = None
    traffic_source = None
    pmc = ROOT / "profiles" / "pmc_summary.json"
    if pmc.exists() and args.config == "cfg3" and n == 8192:          # the committed counter passes are of this workload only
        try:
            pj = json.loads(pmc.read_text())
            traffic = pj.get(names[dom].split("<")[0], {}).get("hbm_bytes_per_launch")
            if traffic is not None:
                traffic_source = ("profiles/pmc_summary.json (committed; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --steps 1 --warmup 3`, the fourth launch after a reset; "
                                  "not collected in this run): " + str(pj.get("_note", ""))[:160])
        except Exception:
            traffic = None
    value = world * n * K / dt
    bad, any_bad = sim.bad_state()
    cap_con, cap_row, cap_item, cap_total = sim.cap_counts()
    nblocks = len(m.free_joint_qadrs())
    dofs = {2: "slide_x/slide_y DOFs only", 7: "all 7 DOFs"}.get(m.nu, f"{m.nu} actuated DOFs")
    baseline_cfg = {"cfg1": "BASELINE config 1 shape", "cfg2": "BASELINE config 2", "cfg3": "BASELINE config 3",
                    "cfg4": "BASELINE configs 4/5 per-GPU shard"}.get(args.config, "SURVEY 8f scene")
    # the roofline that binds: FP32 VALU issue.  Wave-instructions per wave per substep come from the committed
    # PMC pass of this kernel (profiles/pmc_summary.json, SQ_INSTS_VALU); peak = 1024 SIMD-32 x 2.4 GHz / 2 clocks per wave64 op
    valu = None
    try:
        sq = json.loads(pmc.read_text()).get("_sq_per_wave_per_substep", {}) if pmc.exists() else {}
        if persistent and args.config == "cfg3" and "SQ_INSTS_VALU" in sq:
            waves = (n * (16 if m.nv <= 16 else 32) + 63) // 64
            ips = sq["SQ_INSTS_VALU"] * waves * mean_substeps / (avg_us * 1e-6)
            peak_ips = 1024 * 2.4e9 / 2
            valu = {"wave_instr_per_s": ips, "peak": peak_ips, "frac": ips / peak_ips, "flop_per_s_estimate": ips * 64 * 1.3,
                    "source": "profiles/pmc_summary.json _sq_per_wave_per_substep.SQ_INSTS_VALU x waves x substeps / measured launch time; "
                              "flop estimate = lanes x ~1.3 flop per VALU instruction (fma share), most lanes of a 16-lane group idle in the serial phases"}
    except Exception:
        valu = None

    # lifetimes of the workgroups of one launch (the launch ends with its slowest workgroup): measured with the diagnostic build
    # libhsrsim_life.so by tools/block_life.py and committed with the round's profiles
    lifetimes = None
    try:
        lf = ROOT / "profiles" / "block_life.json"
        if lf.exists():
            lifetimes = json.loads(lf.read_text()).get(args.config)
    except Exception:
        lifetimes = None

    out = {
        "metric": f"env-steps/sec (whole node), HSR+{nblocks}-block, steps_per_action={STEPS_PER_ACTION}, {n} envs",
        "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": 1e3 * dt / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{baseline_cfg}: {args.config}, {dofs} + {nblocks} block(s), {n} envs per GPU x {world} GPU(s), "
                               f"steps_per_action={STEPS_PER_ACTION}, geofence={GEOFENCE}, ctrl~U(ctrlrange) per env-step, done envs reset",
                   "envs_per_gpu": n, "global_envs": world * n, "substeps_per_env_step": STEPS_PER_ACTION,
                   "mean_substeps_executed": mean_substeps, "done_fraction": dones / (n * K),
                   "gather_mode": (None if world == 1 else ("overlapped with the next env-step (open loop)" if gather_overlap else "in line (closed loop)")),
                   "parallelism": f"env-shard x{world}" + (f" + all-gather(obs,reward,done) over {'gloo (rehearsal)' if args.rehearse_on_one_gpu else 'RCCL'}, {dist.get_world_size()} ranks{', overlapped with the next env-step' if gather_overlap else ''}" if world > 1 else ""),
                   "cap_hits": {"contacts_beyond_nconmax": cap_con / max(cap_total, 1), "rows_beyond_njmax": cap_row / max(cap_total, 1),
                                "items_beyond_64_per_env": cap_item / max(cap_total, 1), "env_substeps": cap_total,
                                "nconmax_njmax": [int(m.arrays["sizes"][10]), int(m.arrays["sizes"][11])], "note": "fraction of (env, substep) pairs; MuJoCo's own caps are 100 / 500 (world.xml:44)"},
                   "substeps_per_s": value * mean_substeps, "persistent_kernel": persistent, "hipgraph": (not args.no_graph) and not persistent,
                   "bad_envs": int(bad.sum())},
        # bound / achieved / peak / unit / frac / traffic belong together: the HBM report BASELINE prescribes (round-5 advisor: a "valu" bound next to an
        # HBM fraction misleads a consumer that reads the two together); what actually binds - FP32 VALU issue / latency - is `binding` + `valu`
        "roofline": {"bound": "hbm", "kernel": names[dom], "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                     "binding": "valu",
                     "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_us": avg_us, **kernel_stats,
                     "workgroup_lifetimes": lifetimes,
                     "valu": valu,
                     "kernel_ms_per_env_step": {nm: ms for nm, ms in zip(names, k_ms) if nm != "-"},
                     "launches_per_env_step": {nm: k for nm, k in zip(names, k_n) if nm != "-"},
                     "note": "state stays L2/MALL-resident; the path is FP32-VALU/latency bound, not HBM bound (SURVEY.md 8d)"},
    }
    if world == 1 and persistent and not args.no_capacity and args.capacity_envs > n:
        out["capacity"] = capacity_leg(m, args.capacity_envs, dev, local_rank)
    if rank == 0 and not args.no_cpu_baseline:
        # rank 0 only, after the timed region (the other ranks wait at destroy_process_group): a line at N > 1 carries the baseline too
        out["cpu_baseline"] = cpu_baseline(m, q0, goal, ctrl_host, host_cores())
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    sim.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
