"""GPU parity tests aimed at the branches of the hot path that the stage tests do not reach: the persistent kernel's own
narrowphase (compared contact by contact), the ctrl / force clamps with the reference driver's actions, a --set-xml model,
a model that must fall back to the per-substep chain, hipGraph replay of that chain, and the bad-state exception."""
import numpy as np
import pytest

from hsr_env_amd import sim as hs
from oracle.oracle import OracleSim
from test_gpu_parity import contact_mismatch, oracle_rollout, random_states

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfg", ["cfg2", "cfg3", "cfg4", "cupboard", "cfg3_setxml"])
def test_persistent_contacts_match_oracle(models, cfg):
    """The persistent kernel's OWN collision code (geom cache, sphere / box culls, separation-margin cache, 8-lane box-box and
    MPR): after step(ctrl, 1) with the introspection flag the contact records of that substep's forward pass must equal the
    oracle's at the same state: same count per env, depth 1e-5, normal 2e-3, position 2e-4 (the bounds of the stage test of the
    chain kernels).  A contact at the edge of existence (|depth| < 2e-6 in either precision) may appear in one list only: such envs
    are counted and must stay below 2 %; everything else must match exactly.  Three consecutive substeps are checked, the later
    ones from the HIP path's own state, so that the separating-axis / margin caches are exercised fresh and carried over."""
    m = models[cfg]
    n = 96
    rng = np.random.default_rng(110)
    q, v, ctrl = random_states(m, n, rng)
    pre = oracle_rollout(m, q, v, ctrl, 40)
    sim = hs.BatchSim(m, n)
    assert sim.is_persistent()
    sim.set_debug(True)
    sim.set_warmstart(np.array([s.qacc_warmstart for s in pre]))
    sim.set_state(np.zeros(n), np.array([s.qpos for s in pre]), np.array([s.qvel for s in pre]))
    states = [(np.array([s.qpos for s in pre]), np.array([s.qvel for s in pre]))]
    for rnd in range(3):
        sim.step(ctrl, 1)
        ncon = sim.get_field(hs.F_NCON); con = sim.get_field(hs.F_CONTACT)
        qs, vs = states[-1]
        edge, reasons = 0, []
        for e in range(n):
            o = OracleSim(m)
            o.qpos[:] = qs[e]; o.qvel[:] = vs[e]; o.ctrl[:] = ctrl[e]
            o.forward()                                 # contacts of the state the HIP substep started from
            oc = o.contacts()
            gc = con[e][con[e][:, 6] <= 0]
            why = contact_mismatch(m, con[e], oc)
            if why is not None:
                shallow = (len(oc) and np.abs(oc[:, 12]).min() < 2e-6) or (len(gc) and np.abs(gc[:, 6]).min() < 2e-6)
                if why.startswith("count") and shallow:
                    edge += 1
                else:
                    reasons.append((e, why))
            else:
                assert int(ncon[e]) == min(len(oc), int(m.arrays["sizes"][10]))
        # next round: the HIP path's own state after its substep (read back as fp32), margins and separating axes carried over
        t1, q1, v1 = sim.get_state()
        states.append((q1.astype(np.float64), v1.astype(np.float64)))
        print(f"{cfg} round {rnd}: {edge} edge-of-existence envs, {len(reasons)} mismatches")
        assert not reasons, reasons[:5]
        assert edge <= max(1, n // 50)
    assert not sim.bad_state()[1]
    sim.close()


@pytest.mark.parametrize("cfg", ["cfg3", "cfg3_setxml"])
def test_reference_driver_actions_and_clamps(models, cfg):
    """The reference's driver sends zeros plus a mouse delta (hsr/control.py:49-62): zeros are outside arm_lift's ctrlrange
    [2.3, 4.1] and arm_flex's [-1.2, -.5] (world.xml:112,115), +-10 is outside every range, so the ctrl clamp and the force clamp
    of solve_body.inc run.  40 substeps against the oracle from the model's initial state: |dqpos| < 2e-5, |dqvel| < 2e-3 for every
    env.  In cfg3_setxml the arm-lift actuator has no ctrl limit (compiler: ctrllimited=false -> infinite range)."""
    m = models[cfg]
    acts = np.array([np.zeros(m.nu), np.full(m.nu, 10.0), np.full(m.nu, -10.0),
                     np.where(np.arange(m.nu) % 2 == 0, 10.0, -10.0)], dtype=np.float32)
    n = len(acts)
    sim = hs.BatchSim(m, n)
    obs, rew, done, ns = sim.step(acts, 40)
    for e in range(n):
        o = OracleSim(m)
        o.env_step(acts[e].astype(np.float64), 40)
        dq = np.abs(obs[e, :m.nq] - o.qpos).max(); dv = np.abs(obs[e, m.nq:] - o.qvel).max()
        assert dq < 2e-5 and dv < 2e-3, (cfg, e, dq, dv)
    if cfg == "cfg3_setxml":
        # the unlimited arm-lift actuator really is unclamped: from q = 0.3 a ctrl of 1.0 (below the limited range [2.3, 4.1], inside
        # the force range either way) pulls the lift down, while the limited actuator of cfg3 clamps it to 2.3 and pushes up
        q = np.tile(m.qpos0, (2, 1)).astype(np.float32); q[:, 2] = 0.3
        c = np.zeros((2, m.nu), np.float32); c[:, 2] = 1.0; c[:, 3] = -0.8
        res = []
        for mm in (m, models["cfg3"]):
            s2 = hs.BatchSim(mm, 2)
            s2.set_state(np.zeros(2, np.float32), q, np.zeros((2, mm.nv), np.float32))
            res.append(s2.step(c, 40)[0][0, 2])
            s2.close()
        o = OracleSim(m)
        o.qpos[:] = q[0]; o.env_step(c[0].astype(np.float64), 40)
        assert abs(res[0] - o.qpos[2]) < 2e-5 and res[0] < 0.3 - 1e-3 and res[1] > res[0] + 1e-3, (res, o.qpos[2])
    sim.close()


def test_setxml_model_single_substep(models):
    """SURVEY 8f row 2 on hardware: the --set-xml model (other pan friction) through the same per-substep parity as cfg3."""
    m = models["cfg3_setxml"]
    n = 64
    rng = np.random.default_rng(111)
    q, v, ctrl = random_states(models["cfg3"], n, rng)      # cfg3's ranges (the unlimited actuator has none to sample from)
    pre = oracle_rollout(m, q, v, ctrl, 60)
    sim = hs.BatchSim(m, n)
    assert sim.kernel_flags() == 7          # the setters change friction and an actuator limit, not the tree: cfg3's constant instance (scalars + tree tables) serves it
    sim.set_debug(True)
    sim.set_warmstart(np.array([s.qacc_warmstart for s in pre]))
    sim.set_state(np.zeros(n), np.array([s.qpos for s in pre]), np.array([s.qvel for s in pre]))
    obs = sim.step(ctrl, 1)[0]
    ncon = sim.get_field(hs.F_NCON)
    unexplained = []
    for e in range(n):
        o = pre[e]
        o.step()
        dq = np.abs(obs[e, :m.nq] - o.qpos).max()
        dv = (np.abs(obs[e, m.nq:] - o.qvel) / (1 + np.abs(o.qvel))).max()
        if not (dq < 5e-6 and dv < 1e-4) and int(ncon[e]) == o.ncon:
            unexplained.append((e, dq, dv))
    assert not unexplained, unexplained
    sim.close()


def test_per_body_hessian_equals_per_contact_assembly(models, monkeypatch):
    """cfg4 (three blocks): the Newton Hessian of the world <-> free-body contacts is assembled per body (one 6 x 6 matrix per
    block, solve_body.inc body_hess); HSR_NFB=0 keeps the per-contact rank-1 updates.  Same states, one substep: both must agree
    with each other far inside the parity tolerance, and with the oracle as every other single-substep test."""
    m = models["cfg4"]
    n = 128
    rng = np.random.default_rng(2024)
    q, v, ctrl = random_states(m, n, rng)
    pre = oracle_rollout(m, q, v, ctrl, 60)
    outs = []
    for nfb in ("9", "0"):
        monkeypatch.setenv("HSR_NFB", nfb)
        sim = hs.BatchSim(m, n)
        sim.set_debug(True)
        sim.set_warmstart(np.array([s.qacc_warmstart for s in pre]))
        sim.set_state(np.zeros(n), np.array([s.qpos for s in pre]), np.array([s.qvel for s in pre]))
        outs.append((sim.step(ctrl, 1)[0], sim.get_field(hs.F_NCON).copy(), sim.newton_trips().copy()))
        assert not sim.bad_state()[0].any()
        sim.close()
    (a, ncon, ta), (b, _, tb) = outs
    dv = np.abs(a[:, m.nq:] - b[:, m.nq:]) / (1 + np.abs(b[:, m.nq:]))
    assert int((ncon >= 8).sum()) > n // 2                      # the blocks do rest on the table: the per-body path is what ran
    assert dv.max() < 2e-5, dv.max()                             # same minimiser, another summation order
    assert np.abs(ta.astype(int) - tb.astype(int)).max() <= 1   # and the same number of Newton iterations (+-1 at a tolerance tie)
    unexplained = []
    for e in range(n):
        o = pre[e]
        o.step()
        d = (np.abs(a[e, m.nq:] - o.qvel) / (1 + np.abs(o.qvel))).max()
        if not d < 1e-4 and int(ncon[e]) == o.ncon:
            unexplained.append((e, d))
    assert not unexplained, unexplained


def test_sparse_factorisation_equals_the_dense_one(models):
    """cfg4 (robot + three blocks): in an env whose contacts couple at most one block to the robot and no block to another - 99 % of the env-substeps, 95 % of those of
    the hardest tasks - the Newton Hessian is factored with the robot's seven dense steps and then dof t of every block at once (solve_g.h chol_sparse_fwd / chol_sparse_back:
    13 dependent steps and ~400 instructions instead of 25 and ~650); test hook 128 keeps the dense factorisation.  Three batches - blocks apart (the robot touches none or
    one), a finger pushed into a block (one block coupled to the robot), two blocks pushed into each other (dense fallback; the choice is made per WAVE - legal because the two paths are bit-identical - so such an env also
    sends its wave neighbours down the dense path) - one substep and then twenty more: both paths must agree BIT FOR BIT (same arithmetic on the entries that are not structurally zero); the
    single substep of the first batch must follow the oracle like every other single-substep test."""
    m = models["cfg4"]
    n = 128
    rng = np.random.default_rng(2025)
    q, v, ctrl = random_states(m, n, rng)
    pre = oracle_rollout(m, q, v, ctrl, 60)
    qs, vs, ws = np.array([s.qpos for s in pre]), np.array([s.qvel for s in pre]), np.array([s.qacc_warmstart for s in pre])
    a0, a1 = m.free_joint_qadrs()[:2]
    qt = qs.copy()                                   # block 1 overlapping block 0 by 2 mm in every fourth env
    qt[::4, a1:a1 + 3] = qt[::4, a0:a0 + 3] + np.array([0.048, 0.0, 0.0]); qt[::4, a1 + 3:a1 + 7] = qt[::4, a0 + 3:a0 + 7]
    qf = qs.copy()                                   # block 0 dropped between the fingers in every other env (the pinch regime of cfg3, with two more blocks on the table)
    bl, br = m.body_id("hand_l_distal_link"), m.body_id("hand_r_distal_link")
    for e in range(0, n, 2):
        o = OracleSim(m); o.qpos[:] = qf[e]; o.forward()
        qf[e, a0:a0 + 3] = 0.5 * (o.body_xpos(bl) + o.body_xpos(br)) + rng.uniform(-0.01, 0.01, 3)
        quat = rng.normal(size=4); qf[e, a0 + 3:a0 + 7] = quat / np.linalg.norm(quat)
    for tag, qq in (("apart", qs), ("finger on a block", qf), ("blocks touching", qt)):
        outs = []
        for hook in (0, 128):
            sim = hs.BatchSim(m, n)
            sim.set_debug(1 | hook)
            sim.set_warmstart(ws)
            sim.set_state(np.zeros(n), qq, vs)
            one = sim.step(ctrl, 1)[0].copy()
            ncon, trips = sim.get_field(hs.F_NCON).copy(), sim.get_field(hs.F_NITER).copy()
            more = sim.step(ctrl, 20)[0].copy()
            assert not sim.bad_state()[0].any()
            sim.close()
            outs.append((one, more, ncon, trips))
        (a1_, am, ncon, ta), (b1_, bm, _, tb) = outs
        dv = np.abs(a1_[:, m.nq:] - b1_[:, m.nq:]) / (1 + np.abs(b1_[:, m.nq:]))
        dm = np.abs(am - bm).max(axis=1)
        print(f"sparse vs dense ({tag}): one substep max rel dqvel {dv.max():.1e}, after 20 more substeps median |dobs| {np.median(dm):.1e} p90 {np.percentile(dm, 90):.1e} max {dm.max():.1e}; "
              f"Newton iterations differ in {int((ta != tb).sum())} of {n} envs (mean {ta.mean():.1f}); contacts per env mean {ncon.mean():.1f}")
        # the sparse path performs the dense path's arithmetic on the structurally non-zero entries in the same order: BIT-IDENTICAL results - which is what allows the
        # choice to be made per wave (an env's result must never depend on its neighbours)
        assert dv.max() == 0.0 and dm.max() == 0.0 and (ta == tb).all(), (dv.max(), dm.max())
        # ... to the BIT (round-5 advisor: a difference of 0.0 hides -0 against +0 - the dense path adds -0 * x terms that the merged one skips)
        assert np.array_equal(a1_.view(np.uint32), b1_.view(np.uint32)) and np.array_equal(am.view(np.uint32), bm.view(np.uint32))
        if tag == "apart":
            unexplained = []
            for e in range(n):
                o = pre[e]
                o.step()
                d = (np.abs(a1_[e, m.nq:] - o.qvel) / (1 + np.abs(o.qvel))).max()
                if not d < 1e-4 and int(ncon[e]) == o.ncon:
                    unexplained.append((e, d))
            assert not unexplained, unexplained
        elif tag == "blocks touching":
            assert int((ncon[::4] > ncon[1::4]).sum()) > n // 16           # the pushed-together blocks do touch
        else:
            assert int((ncon[::2] != ncon[1::2]).sum()) > n // 8            # the dropped block does meet the fingers


@pytest.mark.parametrize("cfg", ["cfg3", "cfg4"])
@pytest.mark.parametrize("hook", [2, 4])
def test_rarely_taken_solver_branches(models, cfg, hook):
    """Two branches of the Newton solver that ordinary states (almost) never take, forced through hsr_batch_set_debug's test
    hooks: J v per contact (taken when an env's contacts involve more links than the link-velocity scratch holds) and the
    PSD-majorant Hessian with its own line-search start (taken when the exact cone Hessian is not positive definite).  Both reach
    the same minimiser: one substep must match the oracle as in every other single-substep test."""
    m = models[cfg]
    n = 64
    rng = np.random.default_rng(31 + hook)
    q, v, ctrl = random_states(m, n, rng)
    pre = oracle_rollout(m, q, v, ctrl, 60)
    sim = hs.BatchSim(m, n)
    sim.set_debug(1 | hook)
    sim.set_warmstart(np.array([s.qacc_warmstart for s in pre]))
    sim.set_state(np.zeros(n), np.array([s.qpos for s in pre]), np.array([s.qvel for s in pre]))
    obs = sim.step(ctrl, 1)[0]
    ncon = sim.get_field(hs.F_NCON)
    assert not sim.bad_state()[0].any()
    assert int((ncon > 0).sum()) > n // 2
    unexplained = []
    for e in range(n):
        o = pre[e]
        o.step()
        dq = np.abs(obs[e, :m.nq] - o.qpos).max()
        dv = (np.abs(obs[e, m.nq:] - o.qvel) / (1 + np.abs(o.qvel))).max()
        if not (dq < 5e-6 and dv < 1e-4) and int(ncon[e]) == o.ncon:
            unexplained.append((e, dq, dv))
    assert not unexplained, unexplained
    sim.close()


def test_pair_tables_in_global_memory_change_nothing(models, monkeypatch):
    """The cupboard scene (274 candidate pairs) would need 22.4 KB of LDS per workgroup - 7 workgroups per CU, a second round of 256
    workgroups at 8192 envs; its kernel instance reads the packed pair records and geom constants from global memory instead
    (8 per CU).  Same values, same arithmetic: bit-identical with the tables in LDS (HSR_TABLES_GLOBAL=0)."""
    m = models["cupboard"]
    n = 128
    rng = np.random.default_rng(8)
    q, v, ctrl = random_states(m, n, rng)
    outs = []
    for tg in ("1", "0"):
        monkeypatch.setenv("HSR_TABLES_GLOBAL", tg)
        sim = hs.BatchSim(m, n)
        sim.set_state(np.zeros(n), q, v)
        outs.append([sim.step(ctrl, 50)[0].copy() for _ in range(2)])
        sim.close()
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("cfg", ["nv11", "nv23"])
def test_generic_instances_of_the_persistent_kernel(models, cfg):
    """The reference configurations run kernel instances with nv and ndense at compile time; any other model gets the generic
    ones (k_env_step_mf<16, 16, false> / <32, 32, false>: loops to the lane-group size, run-time ndense, per-contact Hessian).
    Five arm dofs + one block (nv 11) and + three blocks (nv 23): one substep against the oracle, then sixty substeps of the
    persistent kernel against the per-substep chain."""
    m = models[cfg]
    n = 64
    rng = np.random.default_rng(404)
    q, v, ctrl = random_states(m, n, rng)
    pre = oracle_rollout(m, q, v, ctrl, 60)
    sim = hs.BatchSim(m, n)
    assert sim.is_persistent()
    sim.set_debug(True)
    sim.set_warmstart(np.array([s.qacc_warmstart for s in pre]))
    sim.set_state(np.zeros(n), np.array([s.qpos for s in pre]), np.array([s.qvel for s in pre]))
    obs = sim.step(ctrl, 1)[0]
    ncon = sim.get_field(hs.F_NCON)
    assert not sim.bad_state()[0].any() and int((ncon > 0).sum()) > n // 2
    unexplained = []
    for e in range(n):
        o = pre[e]
        o.step()
        dq = np.abs(obs[e, :m.nq] - o.qpos).max()
        dv = (np.abs(obs[e, m.nq:] - o.qvel) / (1 + np.abs(o.qvel))).max()
        if not (dq < 5e-6 and dv < 1e-4) and int(ncon[e]) == o.ncon:
            unexplained.append((e, dq, dv))
    assert not unexplained, unexplained
    outs = []
    for persistent in (True, False):
        sim.set_persistent(persistent)
        sim.set_warmstart(np.zeros((n, m.nv)))
        sim.set_state(np.zeros(n), q, np.zeros_like(v))
        outs.append(sim.step(ctrl, 60)[0])
    err = np.abs(outs[0] - outs[1]).max(axis=1)
    assert np.median(err) < 1e-5 and np.percentile(err, 90) < 1e-3, err
    sim.close()


def test_group_sums_are_identical_in_every_lane(models):
    """Regression (round 2): a state of cfg4 (tests/golden/cfg4_lane_uniformity_state.npz, found by replaying the bench) in which
    the lanes of one env disagreed in the last bit of a group sum - the compiler had contracted the product in gsum's argument
    into the first butterfly addition - so that one lane left the line search alone, the broadcasts of the next iteration read
    a disabled lane and the env ended in NaN.  One substep from that state must be finite and match the oracle."""
    import pathlib
    d = np.load(pathlib.Path(__file__).parent / "golden" / "cfg4_lane_uniformity_state.npz")
    m = models["cfg4"]
    n = 8
    sim = hs.BatchSim(m, n)
    q = np.tile(d["qpos"], (n, 1)); v = np.tile(d["qvel"], (n, 1))
    sim.reset(qpos0=q, mocap=np.zeros((n, 3)))
    sim.set_state(np.zeros(n), q, v)
    sim.set_warmstart(np.tile(d["warm"], (n, 1)))
    obs = sim.step(np.tile(d["ctrl"], (n, 1)), 1)[0]
    assert np.isfinite(obs).all() and not sim.bad_state()[0].any()
    assert (obs == obs[0]).all()                                 # the copies share waves pairwise: identical whatever the neighbour
    o = OracleSim(m)
    o.qpos[:] = d["qpos"]; o.qvel[:] = d["qvel"]; o.qacc_warmstart[:] = d["warm"]; o.ctrl[:] = d["ctrl"]
    o.step()
    assert (np.abs(obs[0, m.nq:] - o.qvel) / (1 + np.abs(o.qvel))).max() < 1e-4
    assert np.abs(obs[0, :m.nq] - o.qpos).max() < 5e-6
    sim.close()


def test_model_wider_than_its_lane_group_uses_the_chain(models):
    """nq = 18 > 16 lanes (nv = 16): the persistent kernel holds qpos one entry per lane, so this model must run the
    per-substep chain (which indexes qpos through dof_qposadr) - and match the oracle."""
    m = models["nq18"]
    assert m.nq == 18 and m.nv == 16
    n = 48
    rng = np.random.default_rng(112)
    q, v, ctrl = random_states(m, n, rng)
    sim = hs.BatchSim(m, n)
    assert not sim.is_persistent() and not sim.set_persistent(True)
    pre = oracle_rollout(m, q, v, ctrl, 30)
    sim.set_warmstart(np.array([s.qacc_warmstart for s in pre]))
    sim.set_state(np.zeros(n), np.array([s.qpos for s in pre]), np.array([s.qvel for s in pre]))
    obs = sim.step(ctrl, 1)[0]
    bad = 0
    for e in range(n):
        o = pre[e]
        o.step()
        dq = np.abs(obs[e, :m.nq] - o.qpos).max()
        dv = (np.abs(obs[e, m.nq:] - o.qvel) / (1 + np.abs(o.qvel))).max()
        bad += not (dq < 5e-6 and dv < 1e-4)
    assert bad <= 1, bad
    sim.close()


def test_graph_replay_equals_plain_launches(models):
    """The per-substep chain (fallback path) replayed from its captured hipGraph gives bit-identical results to plain launches."""
    m = models["cfg3"]
    n = 128
    rng = np.random.default_rng(15)
    q, v, ctrl = random_states(m, n, rng)
    outs = []
    for use_graph in (True, False):
        sim = hs.BatchSim(m, n)
        assert sim.set_persistent(False) is False
        sim.set_graph(use_graph)
        sim.set_state(np.zeros(n), q, v)
        o1 = sim.step(ctrl, 50)[0].copy(); o2 = sim.step(ctrl, 50)[0].copy()
        outs.append((o1, o2))
        sim.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


def test_env_raises_on_bad_state(models):
    """mujoco_py raises MujocoException when MuJoCo warns about a diverged state; VecHSREnv.step checks the per-env flags."""
    from hsr_env_amd.env import VecHSREnv
    m = models["cfg2"]
    env = VecHSREnv(model=m, n_envs=8, steps_per_action=5)
    env.reset()
    qpos = np.tile(m.qpos0, (8, 1)); qvel = np.zeros((8, m.nv)); qvel[3, 0] = 1e12
    env.set_state(qpos, qvel)
    with pytest.raises(hs.MujocoException):
        env.step(np.zeros((8, m.nu)))
    env.close()


def test_cap_counters_and_full_size(models):
    """BASELINE size, bench inputs: how often the device buffer caps (16 contacts / 48 rows / 64 items per env) bite - it must be
    (almost) never, otherwise the physics would deviate from nconmax=100 njmax=500 (hsr/models/world.xml:44)."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    from bench import sample_inputs
    m = models["cfg3"]
    n = 8192
    q0, goal = sample_inputs(m, n, 0, 0)
    sim = hs.BatchSim(m, n)
    sim.reset(qpos0=q0, mocap=goal)
    sim.cap_counts()
    rng = np.random.default_rng(1)
    for k in range(2):
        ctrl = rng.uniform(m.act_ctrlrange[:, 0], m.act_ctrlrange[:, 1], (n, m.nu)).astype(np.float32)
        sim.step(ctrl, 300, m.body_id("block0"), 0.05)
    c_con, c_row, c_item, total = sim.cap_counts()
    print(f"cap hits over {total} env-substeps: contacts {c_con}, rows {c_row}, items {c_item}")
    assert total >= 2 * 300 * n * 0.99
    assert c_item == 0 and c_con <= 1e-5 * total and c_row <= 1e-5 * total
    sim.close()


def test_three_blocks_full_size_replay_of_the_bench_stays_finite(models):
    """cfg4 at the bench's size and inputs (8192 envs, 3 blocks, ctrl ~ U(ctrlrange) per env-step, done envs reset), ten env-steps:
    every env finite and not flagged bad, the caps counted over exactly the substeps run, no block through the floor or launched - the run
    that exposed the lane-divergent group sum (env 3704, env-step 7).  The buffer caps may bite at most once in 1e6 env-substeps
    (njmax = 124 rows per env since round 3: 1 row-cap event in 7.4e8 env-substeps of the 300-env-step soak, tools/soak.py; with 96 rows
    it was 4.3e-4)."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    from bench import sample_inputs, GEOFENCE, STEPS_PER_ACTION
    m = models["cfg4"]
    n = 8192
    q0, goal = sample_inputs(m, n, 0, 0)
    rng = np.random.Generator(np.random.Philox(key=[1, 0]))
    lo, hi = m.act_ctrlrange[:, 0].astype(np.float32), m.act_ctrlrange[:, 1].astype(np.float32)
    sim = hs.BatchSim(m, n)
    sim.reset(qpos0=q0, mocap=goal)
    sim.cap_counts()
    bid = m.body_id(m.block_body())
    nsub = 0
    for k in range(10):
        ctrl = rng.uniform(lo, hi, (n, m.nu)).astype(np.float32)
        obs, rew, done, ns = sim.step(ctrl, STEPS_PER_ACTION, bid, GEOFENCE)
        nsub += int(ns.sum())
        assert np.isfinite(obs).all(), (k, np.nonzero(~np.isfinite(obs).all(1))[0][:8])
        assert not sim.bad_state()[0].any(), (k, np.nonzero(sim.bad_state()[0])[0][:8])
        for a in m.free_joint_qadrs():
            z = obs[:, a + 2]
            assert (z > -0.05).all() and (z < 1.0).all(), (k, float(z.min()), float(z.max()))    # pushed off the pan it may fall to the floor; never through it, never launched
            assert np.abs(np.linalg.norm(obs[:, a + 3:a + 7], axis=1) - 1).max() < 1e-4
        rq, rg = sample_inputs(m, n, 2 + k, 0)
        sim.reset(mask=np.asarray(done, np.uint8), qpos0=rq, mocap=rg)
    c_con, c_row, c_item, total = sim.cap_counts()
    assert total == nsub
    assert c_item == 0 and c_con <= 1e-6 * total and c_row <= 1e-6 * total, (c_con, c_row, c_item, total)
    assert sum(sim.cap_histogram()) == c_row
    sim.close()


def test_wave_packing_with_a_ragged_batch(models):
    """The packing sorts 8192 envs per chunk: a batch that is neither a multiple of the chunk nor of the envs per wave (8192 + 13)
    must be covered exactly once - every env stepped, bit-identical with the packing off."""
    m = models["cfg3"]
    n = 8192 + 13
    rng = np.random.default_rng(5)
    q = np.tile(m.qpos0, (n, 1)); a = m.free_joint_qadrs()[0]
    q[:, a] = rng.uniform(-0.1, 0.1, n); q[:, a + 1] = rng.uniform(-0.2, 0.2, n)
    goal = np.tile([0.3, 0.0, 0.422], (n, 1))
    ctrls = [rng.uniform(m.act_ctrlrange[:, 0], m.act_ctrlrange[:, 1], (n, m.nu)).astype(np.float32) for _ in range(3)]
    outs = []
    for schedule in (False, True):
        sim = hs.BatchSim(m, n)
        sim.set_schedule(schedule)
        sim.reset(qpos0=q, mocap=goal)
        res = [sim.step(c, 40, m.body_id("block0"), 0.03) for c in ctrls]
        outs.append(res)
        assert (res[-1][3] == 40).all() or res[-1][2].any()           # every env ran its substeps (or latched done)
        sim.close()
    for k in range(len(ctrls)):
        for x, y in zip(outs[0][k], outs[1][k]):
            assert np.array_equal(x, y), f"packing changed a result in env-step {k}"
    assert not (outs[1][-1][0][:, :m.nq] == q).all(axis=1).any()     # no env was left unstepped


@pytest.mark.parametrize("cfg", ["cfg3", "cfg4"])
def test_wave_packing_never_changes_a_result(models, cfg):
    """The persistent kernel can re-pack the envs over its waves before every launch (hard envs one per wave, easiest envs as
    neighbours).  That is a scheduling decision: with the packing on or off, and with the batch permuted, every env must give
    bit-identical results over several env-steps - no value of an env may ever depend on its wave neighbours (the contact-rich
    random states make sure that envs with many Newton iterations are among them)."""
    m = models[cfg]
    n = 512
    rng = np.random.default_rng(77)
    q, v, ctrl = random_states(m, n, rng)
    goal = np.column_stack([rng.uniform(-0.1, 0.1, n), rng.uniform(-0.2, 0.2, n), np.full(n, 0.422)])
    bid = m.body_id(m.block_body())
    ctrls = [rng.uniform(m.act_ctrlrange[:, 0], m.act_ctrlrange[:, 1], (n, m.nu)).astype(np.float32) for _ in range(3)]
    perm = rng.permutation(n)

    def run(schedule, order):
        sim = hs.BatchSim(m, n)
        sim.set_schedule(schedule)
        sim.reset(qpos0=q[order], mocap=goal[order])
        outs = []
        for c in ctrls:
            obs, rew, done, ns = sim.step(c[order], 100, bid, 0.03)
            outs.append((obs.copy(), done.copy(), ns.copy()))
        sim.close()
        return outs

    ident = np.arange(n)
    ref = run(False, ident)
    on = run(True, ident)
    shuffled = run(True, perm)
    inv = np.argsort(perm)
    for k in range(len(ctrls)):
        for a, b in zip(ref[k], on[k]):
            assert np.array_equal(a, b), f"packing changed a result in env-step {k}"
        for a, b in zip(ref[k], shuffled[k]):
            assert np.array_equal(a, b[inv]), f"the position in the batch changed a result in env-step {k}"


def test_rl_plumbing_on_the_device(models):
    """SURVEY 8f row 4 on hardware: transitions produced by BatchSim.step_dev go into the DeviceReplayBuffer without visiting the
    host (torch tensors on cuda:0, ordered after the step through the batch's stream), and TimeLimit truncates the vectorised env."""
    import torch
    from hsr_env_amd import GoalSpec, VecHSREnv
    from hsr_env_amd.rl import DeviceReplayBuffer, TimeLimit
    m = models["cfg3"]
    n = 64
    dev = torch.device("cuda", 0)
    sim = hs.BatchSim(m, n)
    rng = np.random.default_rng(5)
    q, v, ctrl = random_states(m, n, rng)
    sim.set_state(np.zeros(n), q, v)
    buf = DeviceReplayBuffer(4 * n, device=dev)
    d_obs = torch.empty((n, m.nq + m.nv), dtype=torch.float32, device=dev); d_rew = torch.empty(n, dtype=torch.float32, device=dev)
    d_done = torch.empty(n, dtype=torch.uint8, device=dev); d_ns = torch.empty(n, dtype=torch.int32, device=dev)
    ext = torch.cuda.ExternalStream(sim.stream_ptr(), device=dev)
    host = []
    with torch.cuda.stream(ext):
        for k in range(6):
            d_ctrl = torch.from_numpy(ctrl.astype(np.float32)).to(dev)
            sim.step_dev(d_ctrl.data_ptr(), 10, -1, 0.0, d_obs.data_ptr(), d_rew.data_ptr(), d_done.data_ptr(), d_ns.data_ptr())
            buf.extend((d_obs.clone(), d_rew.clone(), d_done.clone()))
            host.append(d_obs.cpu().numpy().copy())
    torch.cuda.synchronize()
    assert len(buf) == 4 * n and buf.full and buf.buffer[0].device.type == "cuda"
    newest = buf[-n:0][0].cpu().numpy()
    oldest = buf[-4 * n:-3 * n][0].cpu().numpy()
    assert np.array_equal(newest, host[-1]) and np.array_equal(oldest, host[2])
    o, r, d = buf.sample(32)
    assert o.shape == (32, m.nq + m.nv) and o.device.type == "cuda"
    sim.close()
    env = TimeLimit(VecHSREnv(model=models["cfg2"], n_envs=4, goals=[GoalSpec("block0", np.array([.4, 0, .422]), .05)], steps_per_action=3), 2)
    env.reset()
    _, _, done, info = env.step(np.zeros((4, 2)))
    assert not done.any()
    _, _, done, info = env.step(np.zeros((4, 2)))
    assert done.all() and info["TimeLimit.truncated"].all()
    env.close()


def test_several_goals_all_in_range(models):
    """hsr/env.py:124-126: done = all(in_range(*g) for g in goals), tested after every substep.  Two goals - the block within a
    geofence of the sampled point (written to mocap_pos) AND a finger within reach of the block - against the oracle stepped
    substep by substep with the same test in Python; persistent kernel and per-substep chain."""
    from hsr_env_amd import GoalSpec, VecHSREnv
    m = models["cfg3"]
    n = 48
    rng = np.random.default_rng(9)
    q, v, ctrl = random_states(m, n, rng)
    point = q[:, 7:10].copy(); point[n // 2:, 0] += 0.25                 # half of the points at the block, half away
    d_point, d_body = 0.04, 0.75
    bid, fid = m.body_id("block0"), m.body_id("hand_l_distal_link")
    for persistent in (True, False):
        sim = hs.BatchSim(m, n)
        sim.set_persistent(persistent)
        env = VecHSREnv(model=m, n_envs=n, sim=sim, steps_per_action=60,
                        goals=[GoalSpec("block0", np.zeros(3), d_point), GoalSpec("hand_l_distal_link", "block0", d_body)])
        env.reset()
        env._goal_points[:] = point                                      # per-env points (a Box would be sampled; fixed here)
        sim.reset(qpos0=q.astype(np.float32), mocap=point.astype(np.float32))
        obs, rew, done, info = env.step(ctrl)
        ns = info["substeps"]
        for e in range(n):
            o = OracleSim(m)
            o.qpos[:] = q[e]; o.mocap_pos[:] = point[e]; o.ctrl[:] = ctrl[e]
            k, dn = 0, False
            while k < 60 and not dn:
                o.step(); k += 1
                pb, pf = o.body_xpos(bid), o.body_xpos(fid)
                dn = np.linalg.norm(pb - point[e]) < d_point and np.linalg.norm(pf - pb) < d_body
            assert dn == bool(done[e]) and abs(k - int(ns[e])) <= (1 if dn else 0), (persistent, e, k, int(ns[e]), dn, bool(done[e]))
        assert done.any() and not done.all()
        env.close()


@pytest.mark.parametrize("overlap", [False, True])
def test_rccl_all_gather_on_the_batch_stream_one_rank(models, monkeypatch, overlap):
    """The exchange step of the sharded run (hsr_env_amd.dist.StepGather: pack + all_gather_into_tensor, backend "nccl" = RCCL, the
    process group bound to the device, issued on the batch's own HIP stream through torch.cuda.ExternalStream) executed on the one
    GPU there is: a one-rank communicator, so that the first multi-GPU run is not the first execution of this code.  The gathered
    buffer must equal obs | reward | done of the step, with no host synchronisation between the env-step and the collective.
    overlap=True: the collective on StepGather's side stream behind an event, two buffer pairs - the mode bench.py uses for N > 1; the
    buffers of the last TWO env-steps must hold those steps' outputs (the batch stream ran ahead of the collectives)."""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from hsr_env_amd import dist as hd
    m = models["cfg2"]
    n = 256
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    # monkeypatch: the rendezvous variables are gone again after the test (a later test that starts `bench.py --gpus N` must not find WORLD_SIZE=1)
    for k, v in dict(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0").items():
        monkeypatch.setenv(k, v)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    hd.init_process_group("nccl", dev)
    try:
        assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
        rng = np.random.default_rng(5)
        q, v, ctrl = random_states(m, n, rng)
        sim = hs.BatchSim(m, n)
        sim.set_state(np.zeros(n), q, v)
        nobs = m.nq + m.nv
        d_ctrl = torch.from_numpy(ctrl.astype(np.float32)).to(dev)
        d_obs = torch.empty((n, nobs), dtype=torch.float32, device=dev)
        d_rew = torch.empty(n, dtype=torch.float32, device=dev)
        d_done = torch.empty(n, dtype=torch.uint8, device=dev)
        d_ns = torch.empty(n, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        ext = torch.cuda.ExternalStream(sim.stream_ptr(), device=dev)
        with torch.cuda.stream(ext):
            gather = hd.StepGather(n, nobs, 1, dev, always=True, overlap=overlap)
            assert gather.overlap == overlap
            outs, kept = [], []
            for _ in range(4):
                sim.step_dev(d_ctrl.data_ptr(), 25, m.body_id("block0"), 0.5, d_obs.data_ptr(), d_rew.data_ptr(), d_done.data_ptr(), d_ns.data_ptr())
                out = gather(d_obs, d_rew, d_done)
                outs.append(out); kept.append(d_obs.clone())
            gather.wait(out)
        sim.sync()
        torch.cuda.synchronize()
        assert out.data_ptr() not in [p_.data_ptr() for p_ in gather.packs] and out.shape == (n, nobs + 2)
        if overlap:      # two buffer pairs: the last two steps are both still there
            assert outs[-1].data_ptr() != outs[-2].data_ptr() and outs[-1].data_ptr() == outs[-3].data_ptr()
            assert torch.equal(hd.unpack_step(outs[-2])[0], kept[-2]) and not torch.equal(kept[-2], kept[-1])
        o, r, d = hd.unpack_step(out)
        assert torch.equal(o, d_obs) and torch.equal(r, d_rew) and torch.equal(d, d_done > 0)
        t, qq, vv = sim.get_state()
        assert np.array_equal(o.cpu().numpy(), np.concatenate([qq, vv], axis=1))
        assert int(d_ns.min()) >= 1
        sim.close()
    finally:
        dist.destroy_process_group()


def test_step_outputs_written_by_the_persistent_kernel(models):
    """hsr_batch_step_dev with the persistent kernel: ctrl is read and obs / reward / done / nsteps are written by that kernel itself
    (no k_begin_step / k_soa_to_aos / k_end_step launches).  Checked against the state arrays and against the per-substep chain,
    which still goes through those kernels: identical outputs for the same inputs, also for envs that finish early and with
    output pointers left NULL."""
    import torch
    m = models["cfg3"]
    n = 200
    rng = np.random.default_rng(77)
    q, v, ctrl = random_states(m, n, rng)
    goal = np.tile([0.0, 0.0, 0.422], (n, 1)).astype(np.float32)
    outs = []
    for persistent in (True, False):
        sim = hs.BatchSim(m, n)
        sim.set_persistent(persistent)
        assert sim.is_persistent() == persistent
        sim.set_mocap(goal)
        sim.set_state(np.zeros(n), q, v)
        obs, rew, done, ns = sim.step(ctrl, 30, m.body_id("block0"), 0.08)
        t, qq, vv = sim.get_state()
        assert np.array_equal(obs, np.concatenate([qq, vv], axis=1))
        assert np.array_equal(rew, done.astype(np.float32))
        assert ((ns == 30) | (done > 0)).all() and (ns >= 1).all()
        # a second step with every output pointer NULL runs and leaves a consistent state
        d_ctrl = torch.from_numpy(ctrl.astype(np.float32)).cuda()
        sim.step_dev(d_ctrl.data_ptr(), 5, m.body_id("block0"), 0.08, None, None, None, None)
        sim.sync()
        t2, q2, v2 = sim.get_state()
        assert np.isfinite(q2).all() and (t2 >= t).all() and (t2 <= t + 5 * 0.002 + 1e-6).all()
        outs.append((obs, rew, done, ns))
        sim.close()
    assert 0 < outs[0][2].sum() < n, "the case needs both early exits and full env-steps"
    assert np.array_equal(outs[0][2], outs[1][2]) and np.array_equal(outs[0][3], outs[1][3])
    assert np.abs(outs[0][0] - outs[1][0]).max() < 1e-5


@pytest.mark.parametrize("cfg", ["cfg2", "cfg3", "cfg4"])
def test_constant_instances_equal_the_generic_ones(models, cfg, monkeypatch):
    """The reference configurations run kernel instances whose scalar model fields are compile-time constants (cfg_consts.h);
    HSR_NO_CONST=1 runs the same model through the generic instance (fields read from memory).  Same algorithm, same inputs: the
    results agree to rounding (constant folding may evaluate a reciprocal exactly where the hardware instruction is 1 ulp off; round 6: the constant
    instances also compute kinematics, inertia and bias force as straight-line code on their compile-time tree - csrc/kin3.h - where the generic ones walk
    the tree tables: another summation order, same quantities)."""
    m = models[cfg]
    n = 128
    rng = np.random.default_rng(21)
    q, v, ctrl = random_states(m, n, rng)
    outs = []
    for nc in ("0", "1"):
        monkeypatch.setenv("HSR_NO_CONST", nc)
        sim = hs.BatchSim(m, n)
        assert sim.is_persistent()
        assert sim.kernel_flags() == (7 if nc == "0" else 1), sim.kernel_flags()          # constants + compile-time tree (kin3.h) against the generic instance
        sim.set_state(np.zeros(n), q, v)
        outs.append(sim.step(ctrl, 20)[0].copy())
        sim.close()
    d = np.abs(outs[0] - outs[1])
    print(f"{cfg}: constant vs generic instance after 20 substeps: max |d| = {d.max():.2e}, median = {np.median(d):.2e}")
    assert np.percentile(d, 99) < 1e-4 and np.median(d) < 1e-6


@pytest.mark.parametrize("cfg", ["cfg3", "cfg4"])
def test_work_queue_never_changes_a_result(models, cfg):
    """The work queue of the persistent kernel (rounds of a few substeps, workgroups taking (task, round) tickets; automatic when a batch
    has more tasks than the GPU holds workgroups) only decides WHO runs a task WHEN: forced on for a small batch, with rounds of 7
    substeps and an env-step that early exits for some envs, every output and the whole state are bit-identical to the single-pass
    launch, and so is a second env-step from the handed-over state (margins, stamps and warm starts travel through global memory)."""
    m = models[cfg]
    n = 330
    rng = np.random.default_rng(31)
    q, v, ctrl = random_states(m, n, rng)
    goal = np.tile([0.0, 0.0, 0.422], (n, 1)).astype(np.float32)
    res = []
    for mode in (0, 1):
        sim = hs.BatchSim(m, n)
        sim.set_queue(mode, 7)
        sim.set_mocap(goal)
        sim.set_state(np.zeros(n), q, v)
        out = []
        for k in range(2):
            obs, rew, done, ns = sim.step(ctrl, 60, m.body_id("block0"), 0.1)
            t, qq, vv = sim.get_state()
            out += [obs, rew, done, ns, t, qq, vv, sim.get_warmstart()]
        res.append(out)
        sim.close()
    assert 0 < res[0][2].sum() < n, "the case needs early exits and full env-steps"
    for a, b in zip(*res):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("cfg", ["cfg3", "cfg4", "cupboard"])
def test_item_list_never_changes_a_result(models, cfg):
    """Round 4: the persistent kernel keeps its narrowphase item list over several substeps - built from culls that ask "closer than a
    skin of 1 cm" and rebuilt when some geom may have moved half of that (persist.h) - instead of culling every substep.  A pair on the
    list runs its narrowphase, which reports a contact only where there is one; a pair off it cannot touch.  So every output and the
    whole state must be BIT-identical to the run that culls every substep (test hook 64), over env-steps with a driven arm, early
    exits and the work queue's hand-overs."""
    m = models[cfg]
    n = 330
    rng = np.random.default_rng(41)
    q, v, ctrl = random_states(m, n, rng)
    goal = np.tile([0.0, 0.0, 0.422], (n, 1)).astype(np.float32)
    res = []
    for hook in (0, 64):
        sim = hs.BatchSim(m, n)
        sim.set_debug(1 | hook)
        sim.set_queue(1 if cfg == "cfg4" else 0, 25)
        sim.set_mocap(goal)
        sim.set_state(np.zeros(n), q, v)
        out = []
        for k in range(2):
            obs, rew, done, ns = sim.step(ctrl, 150, m.body_id(m.block_body()), 0.1)
            t, qq, vv = sim.get_state()
            out += [obs, rew, done, ns, t, qq, vv, sim.get_warmstart(), sim.get_field(hs.F_NCON)]
        res.append(out)
        assert not sim.bad_state()[1]
        sim.close()
    assert 0 < res[0][2].sum() < n, "the case needs early exits and full env-steps"
    for a, b in zip(*res):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("warm", [True, False])
def test_pinched_block_contacts_follow_the_oracle(models, warm):
    """The regime that sets the launch time: the block between the fingers (here: dropped into them, centimetres of overlap), two to
    five mesh <-> box contacts (MPR) per env that persist, slide and break over the following substeps.  The persistent kernel carries
    pair state from substep to substep there (warm = True, the default: the final portal of a penetrating pair is the starting portal of
    its next run; separating axes and their margins otherwise).  After 1, 8 and 20 further substeps of its own the contacts of the
    next forward pass are compared with the oracle's cold-started fp64 ones at the same state.
    libccd measures the penetration to the final portal TRIANGLE; where the origin projects outside that triangle, depth and direction
    depend on which triangle of the face the run ended on, and fp32 and fp64 do not always end on the same one.  So in this regime a few
    per cent of the envs differ beyond the stage tolerances (depth 1e-5, normal 2e-3, position 2e-4) even from a cold start (measured:
    8 / 1 / 1 of 96 envs at the three checkpoints, 317 / 170 / 92 convex contacts; differences up to 1.3e-3 in depth, 0.16 in the normal).
    The warm start keeps to libccd's path where it can tell that the path matters - a warm run whose witness falls on a triangle edge is
    started over from scratch, and only interior results seed the next run.  What is left: a warm run can end with an interior witness
    (the face's plane) where the cold search ends on a triangle edge.  Measured over the builds of round 3 (the count moves with every
    change of rounding - these states are chaotic): 9-14 envs warm against 8-13 cold at the first checkpoint (of 96 envs, 317 convex
    contacts), 1-4 later - no difference between the modes that the statistic resolves.  The test bounds the share of such envs
    (18 % / 5 % / 5 %) and the size of the differences in both modes, and requires identical contact counts;
    test_pinch_warm_start_against_cold_start_at_the_same_state compares the two modes with each other directly."""
    m = models["cfg3"]
    n = 96
    rng = np.random.default_rng(3)
    q, v, ctrl = random_states(m, n, rng)
    bl, br = m.body_id("hand_l_distal_link"), m.body_id("hand_r_distal_link")
    a = m.free_joint_qadrs()[0]
    fn, g1, g2 = m.arrays["pair_fn"], m.arrays["pair_geom1"], m.arrays["pair_geom2"]
    convex = {(int(g1[p]), int(g2[p])) for p in range(m.npair) if fn[p] == 3}
    for e in range(n):
        o = OracleSim(m); o.qpos[:] = q[e]; o.forward()
        q[e, a:a + 3] = 0.5 * (o.body_xpos(bl) + o.body_xpos(br)) + rng.uniform(-0.01, 0.01, 3)
        quat = rng.normal(size=4); q[e, a + 3:a + 7] = quat / np.linalg.norm(quat)
    v[:] = 0
    sim = hs.BatchSim(m, n)
    assert sim.is_persistent()
    sim.set_mpr_warm(warm)
    sim.set_debug(True)
    sim.set_state(np.zeros(n), q, v)
    nconvex_total = 0
    for gap, share in ((1, 0.18), (8, 0.05), (20, 0.05)):
        sim.step(ctrl, gap)
        t1, q1, v1 = sim.get_state()
        qs, vs = q1.astype(np.float64), v1.astype(np.float64)
        sim.step(ctrl, 1)                                # its forward pass belongs to the state read back above
        con = sim.get_field(hs.F_CONTACT)
        edge, reasons, nconvex = 0, [], 0
        for e in range(n):
            o = OracleSim(m)
            o.qpos[:] = qs[e]; o.qvel[:] = vs[e]; o.ctrl[:] = ctrl[e]
            o.forward()
            oc = o.contacts()
            nconvex += sum((int(r[13]), int(r[14])) in convex for r in oc)
            gc = con[e][con[e][:, 6] <= 0]
            why = contact_mismatch(m, con[e], oc)
            if why is not None:
                shallow = (len(oc) and np.abs(oc[:, 12]).min() < 2e-6) or (len(gc) and np.abs(gc[:, 6]).min() < 2e-6)
                if why.startswith("count") and shallow:
                    edge += 1
                else:
                    reasons.append((e, why))
        nconvex_total += nconvex
        print(f"pinch (warm={warm}), after {gap} more substeps: {nconvex} convex contacts in the oracle, {edge} edge-of-existence envs, "
              f"{len(reasons)} envs beyond the stage tolerances: {reasons[:12]}")
        assert edge <= max(1, n // 50)
        assert len(reasons) <= share * n, reasons
        for e, why in reasons:
            kind, size = why.split()[0], float(why.split()[1])
            assert kind in ("depth", "normal", "position"), (e, why)          # never a different contact count
            assert size < {"depth": 4e-3, "normal": 0.25, "position": 2e-2}[kind], (e, why)
    assert nconvex_total > 150, "the case must exercise mesh <-> box contacts"
    assert not sim.bad_state()[1]
    sim.close()


def test_pinch_allowance_is_precision(models):
    """is the allowance of test_pinched_block_contacts_follow_the_oracle single precision, or a defect?  The same states,
    cold start (no portal carried over), first checkpoint: the HIP contacts against the fp64 oracle AND against the same oracle compiled in single
    precision (oracle/Makefile: libhsr_oracle_f32.so, -DHO_REAL=float; test infrastructure), plus the two oracles against each other - no kernel
    involved.  If ending on another portal triangle in fp32 is what separates kernel and fp64 oracle, the fp32 oracle has to (a) disagree with
    the fp64 one in about as many envs, and (b) the kernel must not be further from the fp64 oracle than the fp32 oracle is."""
    m = models["cfg3"]
    n = 96
    rng = np.random.default_rng(3)
    q, v, ctrl = random_states(m, n, rng)
    bl, br = m.body_id("hand_l_distal_link"), m.body_id("hand_r_distal_link")
    a = m.free_joint_qadrs()[0]
    for e in range(n):
        o = OracleSim(m); o.qpos[:] = q[e]; o.forward()
        q[e, a:a + 3] = 0.5 * (o.body_xpos(bl) + o.body_xpos(br)) + rng.uniform(-0.01, 0.01, 3)
        quat = rng.normal(size=4); q[e, a + 3:a + 7] = quat / np.linalg.norm(quat)
    v[:] = 0
    sim = hs.BatchSim(m, n)
    sim.set_mpr_warm(False)
    sim.set_debug(True)
    sim.set_state(np.zeros(n), q, v)
    sim.step(ctrl, 1)
    t1, q1, v1 = sim.get_state()
    qs, vs = q1.astype(np.float64), v1.astype(np.float64)
    sim.step(ctrl, 1)
    con = sim.get_field(hs.F_CONTACT)
    sim.close()

    def as_slots(oc):          # an oracle contact list in the layout contact_mismatch expects of the HIP path
        slots = np.zeros((m.nslot, 7)); slots[:, 6] = 1.0
        for p in range(m.npair):
            rows = oc[(oc[:, 13] == m.pair_geom1[p]) & (oc[:, 14] == m.pair_geom2[p])] if len(oc) else oc
            for i, r in enumerate(rows[:int(m.pair_slot[p + 1]) - int(m.pair_slot[p])]):
                slots[int(m.pair_slot[p]) + i] = np.r_[r[0:3], r[3:6], r[12]]
        return slots

    hip64, hip32, o32_64 = [], [], []
    for e in range(n):
        lists = []
        for real in ("f64", "f32"):
            o = OracleSim(m, real)
            o.qpos[:] = qs[e]; o.qvel[:] = vs[e]; o.ctrl[:] = ctrl[e]
            o.forward()
            lists.append(o.contacts().astype(np.float64))
        for out, why in ((hip64, contact_mismatch(m, con[e], lists[0])), (hip32, contact_mismatch(m, con[e], lists[1])),
                         (o32_64, contact_mismatch(m, as_slots(lists[1]), lists[0]))):
            if why is not None:
                out.append((e, why))
    print(f"pinch, cold start, first checkpoint, envs beyond the stage tolerances: HIP vs fp64 oracle {len(hip64)}, HIP vs fp32 oracle {len(hip32)}, "
          f"fp32 oracle vs fp64 oracle {len(o32_64)} of {n}\n  HIP/fp64 {hip64}\n  HIP/fp32 {hip32}\n  fp32/fp64 {o32_64}")
    # Measured (round 6, two builds that differ in the rounding of the kinematics - these states are chaotic, the counts move with every such change):
    # HIP vs fp64 oracle 13 / 13 envs, fp32 oracle vs fp64 oracle 13 / 12 - ten or eleven of them the SAME envs with the same differences to three digits
    # (depth 2.3e-5 ... 6.2e-4, normal up to 3e-2: another triangle of the same face) -, HIP vs fp32 oracle 4 / 5 (depth 1.2e-5 ... 4e-4: the fp32 oracle's own,
    # the kernel's - which measures the depth along the portal normal instead of libccd's expanded quadratic -, and an env in which the two fp32 runs end on
    # different triangles).  The allowance of the fp64 comparison is what single precision does to libccd's MPR in this regime, not a defect of the kernels.
    s64, s32 = {e for e, _ in hip64}, {e for e, _ in o32_64}
    assert len(hip32) <= 8 and len(hip32) < len(hip64), (hip32, hip64)                  # the kernel is closer to the fp32 oracle than to the fp64 one
    assert len(s64 - s32) <= 4, ("envs in which the kernel, but not the fp32 oracle, leaves the fp64 oracle", sorted(s64 - s32))
    assert len(o32_64) >= len(hip64) - 4                                                # single precision alone moves as many envs as the kernel does
    for e, why in hip32:
        kind, size = why.split()[0], float(why.split()[1])
        assert kind in ("depth", "normal", "position") and size < {"depth": 4e-3, "normal": 0.25, "position": 2e-2}[kind], (e, why)          # (the bounds of the fp64 comparison)


def test_pinch_warm_start_against_cold_start_at_the_same_state(models):
    """Round-3 advisor: the pinch test bounds each mode against the oracle, so a regression of the warm start of the size of its allowance
    would pass.  Here the two modes meet directly: the warm-started batch runs on (1, 8, 20 substeps, caches carried), a second batch
    is set to ITS state at every checkpoint (set_state voids every cached axis, margin and portal: a cold start) and both compute the
    contacts of the next substep.  Same state, same precision, same code - only the start of MPR differs: the lists must agree to the
    stage tolerances (depth 1e-5, normal 2e-3, position 2e-4, identical counts) except where the cold search ends on an EDGE of its
    final triangle and the warm one inside it (deviations) - those envs are counted and bounded (measured: 2 / 1 / 0 of 96 at the
    three checkpoints, normals apart by up to 0.13; the bound is 5 %)."""
    m = models["cfg3"]
    n = 96
    rng = np.random.default_rng(3)
    q, v, ctrl = random_states(m, n, rng)
    bl, br = m.body_id("hand_l_distal_link"), m.body_id("hand_r_distal_link")
    a = m.free_joint_qadrs()[0]
    for e in range(n):
        o = OracleSim(m); o.qpos[:] = q[e]; o.forward()
        q[e, a:a + 3] = 0.5 * (o.body_xpos(bl) + o.body_xpos(br)) + rng.uniform(-0.01, 0.01, 3)
        quat = rng.normal(size=4); q[e, a + 3:a + 7] = quat / np.linalg.norm(quat)
    warm = hs.BatchSim(m, n); cold = hs.BatchSim(m, n)
    cold.set_mpr_warm(False)
    for s_ in (warm, cold):
        s_.set_debug(True)
    warm.set_state(np.zeros(n), q, np.zeros_like(v))
    slot_pair = np.repeat(np.arange(m.npair), np.diff(m.pair_slot[:m.npair + 1]))
    total = 0
    for gap in (1, 8, 20):
        warm.step(ctrl, gap)
        t1, q1, v1 = warm.get_state(); w1 = warm.get_warmstart()
        cold.set_warmstart(w1); cold.set_state(t1, q1, v1)
        warm.step(ctrl, 1); cold.step(ctrl, 1)
        cw, cc = warm.get_field(hs.F_CONTACT), cold.get_field(hs.F_CONTACT)
        differ, worst = 0, 0.0
        for e in range(n):
            uw, uc = cw[e][:, 6] <= 0, cc[e][:, 6] <= 0
            assert np.array_equal(np.bincount(slot_pair[uw], minlength=m.npair), np.bincount(slot_pair[uc], minlength=m.npair)), (gap, e)
            d = np.abs(cw[e][uw] - cc[e][uc])
            if len(d) and (d[:, 6].max() > 1e-5 or d[:, 3:6].max() > 2e-3 or d[:, 0:3].max() > 2e-4):
                differ += 1; worst = max(worst, float(d[:, 3:6].max()))
            total += int(uw.sum())
        print(f"pinch, warm vs cold start at the same state, after {gap} more substeps: {differ} of {n} envs beyond the stage tolerances (largest normal difference {worst:.3f})")
        assert differ <= 0.05 * n
        # the cold batch has gone one substep further than the state it was given: it is re-seeded at the next checkpoint
    assert total > 300
    warm.close(); cold.close()


def _thrown_blocks(m, n, rng):
    """Blocks thrown at the (static) robot from all around it: positions on a shell around the robot's axis, velocities of 1-3 m/s
    towards it with some scatter, random spin."""
    q = np.tile(m.qpos0, (n, 1)).astype(np.float64)
    v = np.zeros((n, m.nv))
    axis = np.array([-0.46, -0.081])
    ang = rng.uniform(-np.pi, np.pi, n)
    rad = rng.uniform(0.30, 0.45, n)
    q[:, 0] = axis[0] + rad * np.cos(ang); q[:, 1] = axis[1] + rad * np.sin(ang); q[:, 2] = rng.uniform(0.08, 1.0, n)
    quat = rng.normal(size=(n, 4)); q[:, 3:7] = quat / np.linalg.norm(quat, axis=1, keepdims=True)
    speed = rng.uniform(1.0, 3.0, n)
    aim = ang + np.pi + rng.normal(size=n) * 0.5
    v[:, 0] = speed * np.cos(aim); v[:, 1] = speed * np.sin(aim); v[:, 2] = rng.uniform(-1.0, 2.5, n)
    v[:, 3:6] = rng.normal(size=(n, 3)) * 5.0
    return q, v


def test_separation_margins_expire_when_a_pair_was_not_visited(models):
    """the separation margin of a convex pair is only decremented on substeps that visit the pair; a pair that is
    sphere- or box-culled for a while moves unaccounted, and on its return the stale margin made the kernel skip it - contacts missed,
    centimetres of penetration.  Since round 3 a margin carries the env's substep count of its last visit and counts only on the very
    next substep.  The case: 4096 blocks thrown at the static robot (no robot dof: 15 hulls and a cylinder as scenery), bouncing in and
    out of the cull ranges of the hulls for 330 substeps; at fifteen checkpoints the contact count of the persistent kernel's next
    forward pass is compared, env by env, with the per-substep chain kernels at the same state (they re-check the cached axis with
    two support scans on every visit - exact).  With the stamps ignored (test hook 16 = the round-2 behaviour) the same run misses
    contacts; with them it does not.  Round 4: the narrowphase item list is now kept over several substeps, so a pair near contact is
    visited on every one of them and stale margins have become rare (a pair has to leave the list and come back); the arm that shows
    what the stamps prevent therefore also culls every substep (hook 64), and the stamps are checked in both modes."""
    m = models["static1"]
    n = 4096
    missed = {}
    for hook in (0, 64, 16 | 64):
        rng = np.random.default_rng(12)
        q, v = _thrown_blocks(m, n, rng)
        sim = hs.BatchSim(m, n)
        assert sim.is_persistent()
        sim.set_debug(1 | hook)
        ref = hs.BatchSim(m, n)
        ref.set_persistent(False)
        sim.set_state(np.zeros(n), q, v)
        ctrl = np.zeros((n, 0), np.float32)
        tot_missed, tot_extra, tot_con = 0, 0, 0
        for gap in (40,) + (20,) * 14:
            sim.step(ctrl, gap)
            t1, q1, v1 = sim.get_state()
            ok = np.isfinite(q1).all(1) & (np.abs(q1[:, :3]).max(1) < 5.0)
            sim.step(ctrl, 1)                                  # forward pass of the state read back above, margins carried over
            ncon_p = sim.get_field(hs.F_NCON)
            ref.set_state(np.zeros(n), q1, v1)                 # chain kernels: cull + narrowphase from scratch
            ncon_c = ref.get_field(hs.F_NCON)
            tot_missed += int(((ncon_p < ncon_c) & ok).sum()); tot_extra += int(((ncon_p > ncon_c) & ok).sum()); tot_con += int(ncon_c[ok].sum())
        print(f"thrown blocks, stamps {'IGNORED' if hook & 16 else 'on'}, item list {'rebuilt every substep' if hook & 64 else 'kept'}: {tot_con} contacts at the checkpoints, envs with fewer contacts than the chain: {tot_missed}, with more: {tot_extra}")
        missed[hook] = (tot_missed, tot_extra, tot_con)
        sim.close(); ref.close()
    assert missed[0][2] > 300, "the case must produce contacts"
    for hook in (0, 64):
        assert missed[hook][0] == 0 and missed[hook][1] <= 2, missed          # an edge-of-existence contact may differ between the two code paths
    assert missed[16 | 64][0] > 0, "with the stamps ignored the case must expose missed contacts (else it does not test them)"


def test_work_queue_watchdog_flag_is_sticky(models):
    """A launch that the work queue's watchdog drained (persist.h q_claim) must be reported by the NEXT synchronising call, however many
    launches were enqueued in between - the flag is cleared by the host after reading it, never by a launch (round-3 advisor finding:
    k_queue_init used to clear it, so an asynchronous step_dev loop lost a trip as soon as the following launch was enqueued).
    Hook 32 of hsr_batch_set_debug raises the flag after a launch exactly as the watchdog does."""
    import torch
    from hsr_env_amd.sim import DependencyNotInstalled
    m = models["cfg2"]
    n = 128
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(11)
    q, v, ctrl = random_states(m, n, rng)
    sim = hs.BatchSim(m, n)
    sim.set_state(np.zeros(n), q, v)
    sim.set_queue(1, 10)                 # queued launches: k_queue_init runs in front of every one of them
    d_ctrl = torch.from_numpy(ctrl.astype(np.float32)).to(dev)
    d_obs = torch.empty((n, m.nq + m.nv), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    sim.set_debug(32)
    sim.step_dev(d_ctrl.data_ptr(), 40, -1, 0.05, d_obs.data_ptr())        # "drained"
    sim.set_debug(0)
    for _ in range(2):
        sim.step_dev(d_ctrl.data_ptr(), 40, -1, 0.05, d_obs.data_ptr())    # two more launches enqueued behind it
    with pytest.raises(DependencyNotInstalled, match="work-queue"):
        sim.sync()
    sim.sync()                            # reported once, then clear
    # every other synchronising entry point reports it too
    for call in (lambda: sim.kernel_times(), lambda: sim.cap_counts(), lambda: sim.bad_state(), lambda: sim.get_state()):
        sim.set_debug(32)
        sim.step_dev(d_ctrl.data_ptr(), 40, -1, 0.05, d_obs.data_ptr())
        sim.set_debug(0)
        with pytest.raises(DependencyNotInstalled, match="work-queue"):
            call()
        sim.sync()
    sim.close()


@pytest.mark.parametrize("cfg,n", [("cfg1", 1), ("cfg2", 4096), ("cfg3", 8192), ("cfg4", 8192), ("cfg4", 32768), ("cfg4", 65536), ("cupboard", 8192)])
def test_full_size_invariants(models, cfg, n):
    """BASELINE configs 1-3 at their OWN sizes (1 env; 4096 envs x 1 block, slides only; 8192 envs, all DOFs), the per-GPU shard of configs 4 / 5 (8192 envs x 3
    blocks: 4096 two-env tasks through the work queue, the 32-env replay without it), config 4's TOTAL size on one GPU (32768 envs x 3 blocks: 16384 tasks through
    the queue - what its four shards compute, env for env, since an env's result does not depend on the batch it runs in), config 5's TOTAL size likewise (65536 envs x 3
    blocks, 32768 tasks through the queue: what its eight shards compute) and the cupboard scene, on the bench's inputs, two
    env-steps of 300 substeps with the goal test and the reset of finished envs in between, through properties that do not depend on the
    size: every env finite and unflagged, unit quaternions, blocks between floor and ceiling, reward == done and every finished env really
    inside the geofence, 300 substeps run unless finished; the run is deterministic (a second batch on the same inputs is bit-identical); envs
    are independent (32 envs spread over the batch, replayed ALONE in a batch of their own, reproduce their rows bit for bit - so a
    result at this size is the result of the small batches the oracle tests cover); and eight of those envs follow the oracle over the
    first env-step (median |dobs| < 1e-4, as in test_env_step_300_matches_oracle)."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    from bench import sample_inputs, GEOFENCE, STEPS_PER_ACTION
    m = models[cfg]
    q0, goal = sample_inputs(m, n, 0, 0)
    bid = m.body_id(m.block_body()) if m.block_body() else -1
    rng = np.random.Generator(np.random.Philox(key=[1, 0]))
    lo, hi = m.act_ctrlrange[:, 0].astype(np.float32), m.act_ctrlrange[:, 1].astype(np.float32)
    ctrls = [rng.uniform(lo, hi, (n, m.nu)).astype(np.float32) for _ in range(2)]
    resets = [sample_inputs(m, n, 2 + k, 0) for k in range(2)]

    def run(idx):
        sim = hs.BatchSim(m, len(idx))
        sim.reset(qpos0=q0[idx], mocap=goal[idx])
        g = goal[idx].copy()
        outs = []
        for k in range(2):
            obs, rew, done, ns = sim.step(ctrls[k][idx], STEPS_PER_ACTION, bid, GEOFENCE)
            outs.append((obs.copy(), rew.copy(), done.copy(), ns.copy(), g.copy()))
            assert not sim.bad_state()[0].any()
            sim.reset(mask=np.asarray(done, np.uint8), qpos0=resets[k][0][idx], mocap=resets[k][1][idx])
            g[np.asarray(done, bool)] = resets[k][1][idx][np.asarray(done, bool)]
        assert sim.cap_counts()[2] == 0
        sim.close()
        return outs

    everyone = np.arange(n)
    full = run(everyone)
    blocks = m.free_joint_qadrs()
    for obs, rew, done, ns, g in full:
        assert np.isfinite(obs).all()
        assert np.array_equal(rew > 0, np.asarray(done, bool))
        assert ((ns == STEPS_PER_ACTION) | np.asarray(done, bool)).all() and (ns >= 1).all() and (ns <= STEPS_PER_ACTION).all()
        for a in blocks:
            assert np.abs(np.linalg.norm(obs[:, a + 3:a + 7], axis=1) - 1).max() < 1e-4
            assert (obs[:, a + 2] > -0.05).all() and (obs[:, a + 2] < 1.0).all()
        if blocks and np.asarray(done, bool).any():
            d = np.asarray(done, bool)
            a = blocks[0]
            assert (np.linalg.norm(obs[d, a:a + 3] - g[d], axis=1) < GEOFENCE + 1e-5).all()
    again = run(everyone)
    for x, y in zip(full, again):
        for u, w in zip(x[:4], y[:4]):
            assert np.array_equal(u, w)
    sub = np.unique(np.linspace(0, n - 1, min(n, 32)).astype(int))
    alone = run(sub)
    for x, y in zip(full, alone):
        for u, w in zip(x[:4], y[:4]):
            assert np.array_equal(u[sub], w), "an env's result depends on the batch it runs in"
    errs = []
    for e in sub[:: max(1, len(sub) // 8)][:8]:
        o = OracleSim(m)
        o.qpos[:] = q0[e]; o.mocap_pos[:] = goal[e]
        o.env_step(ctrls[0][e].astype(np.float64), STEPS_PER_ACTION, bid, goal[e].astype(np.float64), GEOFENCE)
        if bool(full[0][2][e]) or int(full[0][3][e]) != STEPS_PER_ACTION:
            continue
        errs.append(np.abs(full[0][0][e] - np.concatenate([o.qpos, o.qvel])).max())
    print(f"{cfg} x {n}: |dobs| vs the oracle after the first env-step, {len(errs)} envs: {np.array2string(np.array(errs), precision=1)}")
    assert len(errs) >= 1 and np.median(errs) < 1e-4, errs


@pytest.mark.parametrize("cfg,n", [("cfg3", 8192), ("cfg2", 9001), ("cfg4", 300)])
def test_packing_of_a_launch_follows_its_contract(models, cfg, n):
    """k_schedule (hsrsim.hip): per chunk of 8192 envs the envs are ordered by the Newton iterations of their previous env-step (more first, ties:
    lower index first); the first lane group of task w holds the w-th of that order, the other lane groups are filled from the easy end; every env
    is held exactly once.  Checked against a numpy restatement on the iteration counts the first env-step left behind - a full chunk (every stage of
    the sorting network: in registers, across the lanes of a wave, across waves), two chunks with a ragged second one, and a 32-lane model."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    from bench import sample_inputs, GEOFENCE
    m = models[cfg]
    q0, goal = sample_inputs(m, n, 0, 0)
    bid = m.body_id(m.block_body()) if m.block_body() else -1
    rng = np.random.Generator(np.random.Philox(key=[5, 0]))
    lo, hi = m.act_ctrlrange[:, 0].astype(np.float32), m.act_ctrlrange[:, 1].astype(np.float32)
    sim = hs.BatchSim(m, n)
    assert sim.is_persistent()
    sim.reset(qpos0=q0, mocap=goal)
    for _ in range(3):              # from the reset state nothing touches anything: every env would report one iteration per substep
        sim.step(rng.uniform(lo, hi, (n, m.nu)).astype(np.float32), 300, bid, GEOFENCE)
    trips = sim.newton_trips().astype(np.int64)
    assert trips.max() > trips.min()                       # something to sort by
    sim.step(rng.uniform(lo, hi, (n, m.nu)).astype(np.float32), 3, bid, GEOFENCE)
    epw = 4 if m.nv <= 16 else 2
    got = sim.packing(epw)
    sim.close()
    want = np.full_like(got, -1)
    for e0 in range(0, n, 8192):
        nc = min(8192, n - e0)
        order = e0 + np.lexsort((np.arange(nc), -np.minimum(trips[e0:e0 + nc], 0x1fffe)))        # iterations descending, then index ascending
        nw = (nc + epw - 1) // epw
        for j in range(epw):
            w = np.arange(nw)
            if j == 0:
                idx = w
            else:
                r = (j - 1) * nw + w
                idx = np.where(r < nc - nw, nc - 1 - r, -1)
            ok = (idx >= 0) & (idx < nc)
            want[e0 // epw + w[ok], j] = order[idx[ok]]
    held = np.sort(got[got >= 0])
    assert np.array_equal(held, np.arange(n)), "an env is held twice or not at all"
    assert np.array_equal(got, want), f"{(got != want).sum()} slots differ from the contract"


@pytest.mark.parametrize("cfg", ["cfg2", "cfg3", "cupboard"])
def test_solo_servers_follow_the_plain_run(models, cfg):
    """Round 4: hard envs leave their task at the end of a round of the work queue and a solo server runs them alone in a wave to the end
    of the env-step, the wave's other lane groups as replicas that share out the contacts of the Newton Hessian / gradient (persist.h;
    hsr_batch_set_solo).  With a threshold that every env passes and a server for each env the WHOLE batch is handed over after its
    first round.  The replicas change the summation order of the Hessian and of J^T f, nothing else: against the run without servers the
    env-step bookkeeping (time, substeps run) is identical, the finished envs are the same up to a goal test at the edge, and the states
    agree like two fp32 summation orders do over 90 contact-rich substeps (median |dobs| < 1e-5, 90 % < 2e-3)."""
    m = models[cfg]
    n = 330
    rng = np.random.default_rng(51)
    q, v, ctrl = random_states(m, n, rng)
    goal = np.tile([0.0, 0.0, 0.422], (n, 1)).astype(np.float32)
    res = []
    for servers in (0, n):
        sim = hs.BatchSim(m, n)
        assert sim.set_solo(servers, 0.01)
        sim.set_queue(1, 10)
        sim.set_mocap(goal)
        sim.set_state(np.zeros(n), q, v)
        obs, rew, done, ns = sim.step(ctrl, 90, m.body_id(m.block_body()), 0.1)
        t, qq, vv = sim.get_state()
        res.append((obs, rew, np.asarray(done, bool), ns, t))
        assert not sim.bad_state()[1]
        if servers:
            ho = sim.solo_handovers()
            assert ho >= (~res[0][2]).sum() * 0.9, f"{ho} envs handed over: the case must hand (nearly) every unfinished env over"
        sim.close()
    (o0, r0, d0, n0, t0), (o1, r1, d1, n1, t1) = res
    assert 0 < d0.sum() < n, "the case needs early exits and full env-steps"
    same = (d0 == d1) & (n0 == n1)
    assert same.mean() >= 0.99, f"{(~same).sum()} envs finish differently"
    assert np.array_equal(t0[same], t1[same]) and np.array_equal(r0[same], r1[same])
    err = np.abs(o0[same] - o1[same]).max(1)
    print(f"{cfg}: solo servers vs plain run over 90 substeps: |dobs| median {np.median(err):.2e}, 90 % {np.percentile(err, 90):.2e}, max {err.max():.2e}; {(~same).sum()} envs finish differently")
    assert np.median(err) < 1e-5 and np.percentile(err, 90) < 2e-3


def test_solo_servers_stand_down_when_the_ticket_cannot_hold_the_substep(models):
    """Round-4 advisor finding: a hand-over ticket packs `env | substep << 20` into one int that the server reads as "empty" when negative - with 2048 or more
    substeps the substep field reaches the sign bit and an env handed over would never be run to the end.  hsr_batch_step now launches such an env-step without
    servers: with servers requested and a threshold every env passes, 2100 substeps must complete (no hand-over, every env stepped 2100 times) and equal, bit
    for bit, the run of a batch that never asked for servers (same queue settings)."""
    m = models["cfg3"]
    n = 64
    rng = np.random.default_rng(52)
    q, v, ctrl = random_states(m, n, rng)
    outs = []
    for servers in (0, n):
        sim = hs.BatchSim(m, n)
        if servers:
            assert sim.set_solo(servers, 0.01)
        sim.set_queue(1, 10)
        sim.set_state(np.zeros(n), q, v)
        obs, rew, done, ns = sim.step(ctrl, 2100)
        assert (ns == 2100).all() and not sim.bad_state()[1]
        if servers:
            assert sim.solo_handovers() == 0
        outs.append(obs.copy())
        sim.close()
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("cfg", ["meshrest4", "meshrest1"])
def test_plane_convex_contacts_match_oracle(models, cfg):
    """Plane <-> convex (round 4: several points per pair in meshrest4, the deepest one only in meshrest1): the head-pan hull dropped on the
    floor plane from 64 small tilts and heights; after the oracle has let it land (120 substeps: some lie on the face, some stand on an
    edge, some are in the air) the HIP path's contacts of the next substep must equal the oracle's contact for contact (depth 1e-5, normal
    2e-3, position 2e-4; a contact with |depth| < 2e-6 may exist in one precision only) and the substep must land on the oracle's
    (|dqpos| < 5e-6, |dqvel| < 1e-4 (1 + |qvel|)) wherever the contact lists agree."""
    m = models[cfg]
    n = 64
    rng = np.random.default_rng(61)
    q = np.tile(m.qpos0, (n, 1))
    q[:, 2] += rng.uniform(0.0, 0.03, n)
    ax = rng.normal(size=(n, 3)); ax /= np.linalg.norm(ax, axis=1, keepdims=True)
    ang = rng.uniform(0, 0.35, n)
    q[:, 3] = np.cos(ang / 2); q[:, 4:7] = ax * np.sin(ang / 2)[:, None]
    v = np.zeros((n, m.nv)); v[:, 3:6] = rng.normal(size=(n, 3)) * 0.3
    ctrl = np.zeros((n, m.nu))
    pre = oracle_rollout(m, q, v, ctrl, 120)
    q1 = np.array([s.qpos for s in pre]); v1 = np.array([s.qvel for s in pre]); w1 = np.array([s.qacc_warmstart for s in pre])
    sim = hs.BatchSim(m, n)
    sim.set_debug(True)
    sim.set_warmstart(w1)
    sim.set_state(np.zeros(n), q1, v1)
    obs = sim.step(ctrl, 1)[0]
    con = sim.get_field(hs.F_CONTACT)
    edge, multi, touching = 0, 0, 0
    for e in range(n):
        o = OracleSim(m)
        o.qpos[:] = q1[e]; o.qvel[:] = v1[e]; o.qacc_warmstart[:] = w1[e]
        o.set_euler_rhs(True)
        o.step()
        oc = o.contacts()
        touching += len(oc) > 0; multi += len(oc) > 1
        why = contact_mismatch(m, con[e], oc)
        if why is not None:
            gc = con[e][con[e][:, 6] <= 0]
            shallow = (len(oc) and np.abs(oc[:, 12]).min() < 2e-6) or (len(gc) and np.abs(gc[:, 6]).min() < 2e-6)
            assert why.startswith("count") and shallow, (e, why)
            edge += 1
            continue
        dq = np.abs(obs[e, :m.nq] - o.qpos).max(); dv = (np.abs(obs[e, m.nq:] - o.qvel) / (1 + np.abs(o.qvel))).max()
        assert dq < 5e-6 and dv < 1e-4, (e, dq, dv)
    print(f"{cfg}: {touching} of {n} hulls touch the floor, {multi} with more than one contact, {edge} edge-of-existence envs")
    assert touching >= n // 2 and edge <= 2
    assert (multi >= n // 4) if cfg == "meshrest4" else (multi == 0)
    sim.close()


def test_hull_of_256_vertices_keeps_its_portal_warm_start(models):
    """Round-3 advisor: the portal warm start of MPR packed its vertex ids in 8 bits with 0xff as "no vertex".  The ids now take 12 bits
    per shape (hulls are refused beyond 256 vertices: collide.h MAXMESHV).  The case: the left finger tip's hull re-listed with 224
    interior points IN FRONT of its 32 vertices - the same convex body, but every vertex that can support now has an id in 224 .. 255,
    the last one the old "no vertex" marker.  The pinch scene (blocks dropped between the fingers: penetrating mesh <-> box pairs that
    persist) must come out bit for bit as with the original vertex list, warm start on, and with contacts on that finger."""
    from hsr_env_amd.compiler import Model, SZ_NMESHVERT
    m0 = models["cfg3"]
    g = list(m0.names["geom"]).index("hand_l_distal_link:l_distal")
    arrays = {k: np.array(v, copy=True) for k, v in m0.arrays.items()}
    mv = arrays["mesh_vert"].reshape(-1, 3)
    a0, n0 = int(arrays["geom_meshadr"][g]), int(arrays["geom_meshnum"][g])
    hull = mv[a0:a0 + n0]
    cen = hull.mean(0)
    inner = np.concatenate([cen + s * (hull - cen) for s in (0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8)])[:256 - n0]
    assert len(inner) == 256 - n0
    arrays["mesh_vert"] = np.concatenate([mv, inner, hull]).reshape((-1, 3) if m0.arrays["mesh_vert"].ndim == 2 else (-1,))
    arrays["geom_meshadr"][g] = len(mv); arrays["geom_meshnum"][g] = 256
    arrays["sizes"][SZ_NMESHVERT] = len(mv) + 256
    m1 = Model(arrays=arrays, names=m0.names, meta=m0.meta)
    m1 = Model.from_bytes(m1.to_bytes())
    n = 96
    rng = np.random.default_rng(3)
    q, v, ctrl = random_states(m0, n, rng)
    bl, br = m0.body_id("hand_l_distal_link"), m0.body_id("hand_r_distal_link")
    a = m0.free_joint_qadrs()[0]
    for e in range(n):
        o = OracleSim(m0); o.qpos[:] = q[e]; o.forward()
        q[e, a:a + 3] = 0.5 * (o.body_xpos(bl) + o.body_xpos(br)) + rng.uniform(-0.01, 0.01, 3)
        quat = rng.normal(size=4); q[e, a + 3:a + 7] = quat / np.linalg.norm(quat)
    res = []
    for m in (m0, m1):
        sim = hs.BatchSim(m, n)
        sim.set_debug(True)
        sim.set_state(np.zeros(n), q, np.zeros_like(v))
        out = []
        for gap in (1, 8, 20):
            obs = sim.step(ctrl, gap)[0]
            out += [obs, sim.get_field(hs.F_CONTACT), sim.get_field(hs.F_NCON)]
        res.append(out)
        assert not sim.bad_state()[1]
        sim.close()
    p_finger = [p for p in range(m0.npair) if int(m0.arrays["pair_geom1"][p]) == g or int(m0.arrays["pair_geom2"][p]) == g]
    slots = np.concatenate([np.arange(m0.pair_slot[p], m0.pair_slot[p + 1]) for p in p_finger])
    assert (res[0][1][:, slots, 6] <= 0).sum() >= 10, "the case needs contacts on the re-listed hull"
    for x, y in zip(*res):
        assert np.array_equal(x, y)


def test_solver_optimum_on_hard_states(models):
    """VERDICT round 4, item 3, GPU half: at the 250 hard solver states of tests/golden/solver_states.npz (pinch regime, the cupboard substeps of the
    round-4 line-search outliers, the bench's regime, three blocks) HSR_F_QACC of one substep through the C-ABI is compared with the MINIMISER of the
    constraint cost - found by scipy on a numpy restatement of the cost (tests/test_oracle_optimality.py: neither the oracle's cone routines nor its
    Newton solver are involved; the oracle only supplies M, J, aref, R of its forward pass at the same state).  States whose fp32 contact list differs
    from the fp64 one beyond the stage tolerances pose a different problem and are set aside (counted, bounded).  fp32 bounds: the scaled cost of the
    HIP solution lies within 1e-6 (1 + |scaled cost|) of the minimum (measured: median 9e-10, worst 3.8e-8; fp32 resolves the cost itself to 6e-8 of its value), |qacc - a*| < 1.1 (2e-2 + 2e-3 |a*|) per dof in EVERY kept state (measured worst 0.98), at most 5 % of the states set aside (measured 8 of 250), at least 30 three-block states kept; formerly: in 95 % of the states (the tolerance of the substep test)."""
    import test_oracle_optimality as too
    rows = list(too.load_states())
    by_cfg = {}
    for r in rows:
        by_cfg.setdefault(id(r[2]), []).append(r)          # load_states hands out one model object per configuration
    excess, rel, skipped, total, kept_nv25 = [], [], 0, 0, 0
    for key, rs in by_cfg.items():
        m = rs[0][2]
        n = len(rs)
        sim = hs.BatchSim(m, n)
        assert sim.is_persistent()
        sim.set_debug(True)
        sim.set_warmstart(np.array([r[5] for r in rs]))
        sim.set_state(np.zeros(n), np.array([r[3] for r in rs]), np.array([r[4] for r in rs]))
        sim.step(np.array([r[6] for r in rs]), 1)
        qacc = sim.get_field(hs.F_QACC).astype(np.float64); con = sim.get_field(hs.F_CONTACT)
        assert not sim.bad_state()[1]
        sim.close()
        for k, (i, rg, m_, q, v, w, c) in enumerate(rs):
            # the problem as fp32 inputs pose it: the HIP path rounds qpos / qvel / warm start to fp32 before anything else
            o = too.oracle_at(m, q.astype(np.float32).astype(np.float64), v.astype(np.float32).astype(np.float64), w.astype(np.float32).astype(np.float64), c.astype(np.float32).astype(np.float64))
            total += 1
            if contact_mismatch(m, con[k], o.contacts()) is not None:
                skipped += 1
                continue
            P = too.Problem(o)
            b = too.minimise(P, P.qas.copy())
            excess.append(P.scale * (P.cost(qacc[k]) - P.cost(b)) / (1.0 + P.scale * abs(P.cost(b))))
            rel.append(float(np.max(np.abs(qacc[k] - b) / (2e-2 + 2e-3 * np.abs(b)))))
            kept_nv25 += m.nv == 25
    excess, rel = np.array(excess), np.array(rel)
    print(f"solver optimum on hard states: {total} states, {skipped} with a contact list that differs from the fp64 one; scaled cost above the minimum / (1 + |scaled cost|): "
          f"median {np.median(excess):.1e} p90 {np.percentile(excess, 90):.1e} max {excess.max():.1e}; |qacc - a*| / (2e-2 + 2e-3 |a*|): median {np.median(rel):.2f} "
          f"p95 {np.percentile(rel, 95):.2f} max {rel.max():.2f}")
    # the bounds are the measurements with a margin (round-5 advisor): 8 of 250 states set aside, every kept state inside the per-dof bound (worst 0.98),
    # and the three-block states (the sparse factorisation of the 32-lane instance) must be among the kept ones
    assert total >= 200 and skipped <= 0.05 * total, (skipped, total)
    assert kept_nv25 >= 30, kept_nv25
    assert excess.max() < 1e-6, excess.max()
    assert rel.max() < 1.1, np.sort(rel)[-10:]
