"""GPU parity tests: the HIP path (through the C-ABI) against the CPU fp64 oracle on the same
seeded inputs.  Stated fp32 tolerances are at the top of each test."""
import numpy as np
import pytest

from hsr_env_amd import sim as hs
from oracle.oracle import OracleSim

pytestmark = pytest.mark.gpu


def random_states(m, n, rng, settle=True):
    """Seeded per-env states: block(s) on the pan (xy in [-.1,.1]x[-.2,.2], yaw uniform; ranges from
    hsr/__init__.py:16-17), robot dofs inside their ranges, small velocities."""
    q = np.tile(m.qpos0, (n, 1))
    v = np.zeros((n, m.nv))
    qa, da = m.scalar_joints()
    nrob = len(qa)
    lo = np.where(m.dof_limited[da] > 0, m.dof_range[da, 0], -1.0)
    hi = np.where(m.dof_limited[da] > 0, m.dof_range[da, 1], -0.4)
    q[:, qa] = rng.uniform(lo, hi, (n, nrob))
    v[:, da] = rng.normal(size=(n, nrob)) * 0.05
    blocks = m.free_joint_qadrs()
    nb = len(blocks)
    for b, a in enumerate(blocks):
        yaw = rng.uniform(-np.pi, np.pi, n)
        q[:, a] = rng.uniform(-0.1, 0.1, n)
        q[:, a + 1] = rng.uniform(-0.2, 0.2, n) if nb == 1 else rng.uniform(-0.05, 0.05, n) + 0.12 * (b - (nb - 1) / 2)
        q[:, a + 2] = 0.422
        q[:, a + 3] = np.cos(yaw / 2); q[:, a + 4:a + 6] = 0; q[:, a + 6] = np.sin(yaw / 2)
    if "cupboard" in m.names["body"]:
        # the uniformly sampled arm can end up deep inside the cupboard walls (centimetres of interpenetration, more
        # contacts than the buffers hold): keep physically plausible states only - resample the robot until no contact is
        # deeper than 3 mm (judged by the oracle's forward pass), falling back to the retracted pose
        for e in range(n):
            for attempt in range(40):
                o = OracleSim(m)
                o.qpos[:] = q[e]
                o.forward()
                c = o.contacts()
                if len(c) <= 8 and (len(c) == 0 or c[:, 12].min() > -3e-3):
                    break
                q[e, qa] = rng.uniform(lo, hi, nrob) if attempt < 39 else m.qpos0[qa]
    ctrl = rng.uniform(m.act_ctrlrange[:, 0], m.act_ctrlrange[:, 1], (n, m.nu)) if m.nu else np.zeros((n, 0))
    return q, v, ctrl


def oracle_rollout(m, q, v, ctrl, nsteps):
    """Advance every env with the oracle; returns per-env OracleSim objects (state after nsteps)."""
    sims = []
    for e in range(q.shape[0]):
        s = OracleSim(m)
        s.qpos[:] = q[e]; s.qvel[:] = v[e]; s.ctrl[:] = ctrl[e]
        for _ in range(nsteps):
            s.step()
        sims.append(s)
    return sims


# envs of the forward-stage test whose contact COUNT differs from the oracle's (a contact at the edge of existence in one precision):
# their kinematics, inertia and smooth acceleration are compared like everyone's, only the contact records and the constrained
# acceleration are not - and how many there may be is pinned per configuration to what was measured (round 4), not a blanket share
FORWARD_NCON_MISMATCH = {"cfg1": 0, "cfg2": 0, "cfg3": 0, "cfg4": 0, "cupboard": 0}


@pytest.mark.parametrize("cfg", ["cfg1", "cfg2", "cfg3", "cfg4", "cupboard"])
def test_forward_stages_match_oracle(models, cfg):
    """sim.forward(): kinematics (|dx| < 2e-6), inertia (rel 1e-5), smooth acceleration (rel 2e-4) for EVERY env;
    contact set (same count; pos/normal/dist 1e-4) and constrained acceleration (abs 2e-2 + rel 2e-3) for every env whose contact
    count equals the oracle's - the others are counted and the count is pinned (FORWARD_NCON_MISMATCH)."""
    m = models[cfg]
    n = 96
    rng = np.random.default_rng(10)
    q, v, ctrl = random_states(m, n, rng)
    # pre-roll with the oracle so that realistic contact states (settled blocks, driven arm) appear
    pre = oracle_rollout(m, q, v, ctrl, 40)
    q = np.array([s.qpos for s in pre]); v = np.array([s.qvel for s in pre]); w = np.array([s.qacc_warmstart for s in pre])
    sim = hs.BatchSim(m, n)
    sim.set_warmstart(w)
    sim.set_state(np.zeros(n), q, v)
    # ctrl enters through step(); forward uses the stored ctrl -> run a 0-substep step to load it
    sim.step(ctrl, 0)
    sim.forward()
    xpos = sim.get_field(hs.F_XPOS); M = sim.get_field(hs.F_M)
    qas = sim.get_field(hs.F_QACC_SMOOTH); qacc = sim.get_field(hs.F_QACC)
    ncon = sim.get_field(hs.F_NCON); con = sim.get_field(hs.F_CONTACT)
    nmis = 0
    for e in range(n):
        o = OracleSim(m)
        o.qpos[:] = q[e]; o.qvel[:] = v[e]; o.ctrl[:] = ctrl[e]; o.qacc_warmstart[:] = w[e]
        o.forward()
        assert np.abs(xpos[e] - o.xpos).max() < 2e-6
        assert np.allclose(M[e], o.M, rtol=1e-5, atol=1e-6)
        assert np.allclose(qas[e], o.qacc_smooth, rtol=2e-4, atol=2e-3)
        if int(ncon[e]) != o.ncon:
            nmis += 1
            continue
        oc = o.contacts()
        gc = con[e][con[e][:, 6] <= 0]
        assert len(gc) == len(oc)
        assert np.allclose(gc[:, 6], oc[:, 12], atol=1e-5)
        assert np.allclose(gc[:, 3:6], oc[:, 3:6], atol=2e-3)
        assert np.allclose(gc[:, 0:3], oc[:, 0:3], atol=2e-4)
        assert np.allclose(qacc[e], o.qacc, rtol=2e-3, atol=2e-2), (e, qacc[e], o.qacc)
    print(f"{cfg}: {nmis} of {n} envs with a contact count different from the oracle's (pinned: {FORWARD_NCON_MISMATCH[cfg]})")
    assert nmis <= FORWARD_NCON_MISMATCH[cfg], f"{nmis} envs with a different contact count"
    sim.close()


# envs of the single-substep test that may sit outside the tolerance BECAUSE a contact exists in one precision only (HIP contact count
# != the oracle's; round 2 allowed 2 % of the envs): none has ever been observed on any configuration (rounds 2 and 3: 0 of 128 on
# all five), so none is allowed; an observed one would have to be entered here with its explanation
EXCUSED_ENVS = {"cfg1": 0, "cfg2": 0, "cfg3": 0, "cfg4": 0, "cupboard": 0}


@pytest.mark.parametrize("cfg", ["cfg1", "cfg2", "cfg3", "cfg4", "cupboard"])
def test_single_substep_matches_oracle(models, cfg):
    """One substep from identical states through the persistent kernel: |dqpos| < 5e-6, |dqvel| < 1e-4 (1 + |qvel|) against the
    oracle's mj_Euler in MuJoCo's form (right-hand side qfrc_smooth + qfrc_constraint).  The HIP path integrates M qacc instead
    (identical at the solver's fixed point; DESIGN.md, deviations): the same substep is therefore also compared like for like, with the
    oracle's Euler switched to M qacc, and both distances are printed - the tolerance must hold for BOTH.  An env outside it must be
    EXPLAINED by a contact that exists in only one precision (HIP contact count of that substep != the oracle's), and how many such
    envs a configuration may have is fixed per configuration (EXCUSED_ENVS: none on cfg1-3); an unexplained env fails the test."""
    m = models[cfg]
    n = 128
    rng = np.random.default_rng(11)
    q, v, ctrl = random_states(m, n, rng)
    pre = oracle_rollout(m, q, v, ctrl, 60)
    q = np.array([s.qpos for s in pre]); v = np.array([s.qvel for s in pre]); w = np.array([s.qacc_warmstart for s in pre])
    sim = hs.BatchSim(m, n)
    sim.set_debug(True)
    sim.set_warmstart(w)
    sim.set_state(np.zeros(n), q, v)
    obs, rew, done, ns = sim.step(ctrl, 1)
    ncon = sim.get_field(hs.F_NCON)
    assert (ns == 1).all() and not done.any()
    explained, unexplained = [], []
    worst = {"force": [0.0, 0.0], "m_qacc": [0.0, 0.0]}
    for e in range(n):
        for form in ("m_qacc", "force"):
            o = OracleSim(m)
            o.qpos[:] = q[e]; o.qvel[:] = v[e]; o.ctrl[:] = ctrl[e]; o.qacc_warmstart[:] = w[e]
            o.set_euler_rhs(form == "m_qacc")
            o.step()
            dq = np.abs(obs[e, :m.nq] - o.qpos).max()
            dv = (np.abs(obs[e, m.nq:] - o.qvel) / (1 + np.abs(o.qvel))).max()
            same_contacts = int(ncon[e]) == o.ncon
            if same_contacts:
                worst[form][0] = max(worst[form][0], float(dq)); worst[form][1] = max(worst[form][1], float(dv))
            if form == "force" and not (dq < 5e-6 and dv < 1e-4):
                (unexplained if same_contacts else explained).append((e, float(dq), float(dv), int(ncon[e]), o.ncon))
            if form == "m_qacc" and same_contacts:
                assert dq < 5e-6 and dv < 1e-4, (e, dq, dv)
    print(f"{cfg}: {len(explained)} of {n} envs outside tolerance with a different contact count (allowed {EXCUSED_ENVS[cfg]}), {len(unexplained)} unexplained; "
          f"worst env with equal contacts: vs MuJoCo's Euler |dqpos| {worst['force'][0]:.2e} |dqvel|/(1+|qvel|) {worst['force'][1]:.2e}, "
          f"vs Euler on M qacc {worst['m_qacc'][0]:.2e} {worst['m_qacc'][1]:.2e}")
    assert not unexplained, unexplained[:5]
    assert len(explained) <= EXCUSED_ENVS[cfg], explained[:5]
    assert not sim.bad_state()[1]
    sim.close()


def contact_mismatch(m, slots, oc):
    """None when the HIP contact records slots[nslot,7] (pos3 normal3 dist; dist = +1: empty) equal the oracle's oc[k,17] to fp32
    tolerance, else a reason.  Contacts are matched pair by pair; inside a pair (box-box: up to 8 points) the order of the points
    is free - the 8-lane clipper emits the polygon from another starting vertex than the sequential one."""
    used = slots[:, 6] <= 0
    if int(used.sum()) != len(oc):
        return f"count {int(used.sum())} vs {len(oc)}"
    for p in range(m.npair):
        a, b = int(m.pair_slot[p]), int(m.pair_slot[p + 1])
        gp = slots[a:b][used[a:b]]
        op = oc[(oc[:, 13] == m.pair_geom1[p]) & (oc[:, 14] == m.pair_geom2[p])]
        if len(gp) != len(op):
            return f"count of pair {p}: {len(gp)} vs {len(op)}"
        left = list(range(len(op)))
        for g in gp:
            k = min(left, key=lambda i: np.abs(op[i, 0:3] - g[0:3]).max())
            left.remove(k)
            if abs(g[6] - op[k, 12]) > 1e-5:
                return f"depth {abs(g[6] - op[k, 12]):.2e} (pair {p})"
            if np.abs(g[3:6] - op[k, 3:6]).max() > 2e-3:
                return f"normal {np.abs(g[3:6] - op[k, 3:6]).max():.2e} (pair {p})"
            if np.abs(g[0:3] - op[k, 0:3]).max() > 2e-4:
                return f"position {np.abs(g[0:3] - op[k, 0:3]).max():.2e} (pair {p})"
    return None


def first_contact_divergence(m, q0, ctrl, nsub):
    """Replays env-steps substep by substep on the GPU (one launch per substep, contact records stored) and in the oracle; returns per env
    the first substep at which the two contact LISTS differ beyond the stage tolerances (contact_mismatch: the number of contacts of some
    candidate geom pair, or a depth / normal / position: -1 = never), the largest |dobs| seen before that substep, and what differed."""
    n = q0.shape[0]
    sim = hs.BatchSim(m, n)
    sim.set_debug(True)
    sim.set_state(np.zeros(n), q0, np.zeros((n, m.nv)))
    orc = []
    for e in range(n):
        o = OracleSim(m)
        o.qpos[:] = q0[e]; o.ctrl[:] = ctrl[e]
        orc.append(o)
    first = np.full(n, -1); before = np.zeros(n); reason = [None] * n; big = [None] * n
    for k in range(nsub):
        obs = sim.step(ctrl, 1)[0]
        con = sim.get_field(hs.F_CONTACT)
        for e, o in enumerate(orc):
            if first[e] >= 0:
                continue
            o.step()
            why = contact_mismatch(m, con[e], o.contacts())
            if why is not None:
                reason[e] = why
                first[e] = k
            else:
                dd = np.abs(obs[e] - np.concatenate([o.qpos, o.qvel]))
                before[e] = max(before[e], float(dd.max()))
                if big[e] is None and dd.max() >= 2e-3:
                    big[e] = (k, int(dd.argmax()), float(dd.max()), int(sim.get_field(hs.F_NITER)[e]), int(o.solver_niter))
        if (first >= 0).all():
            break
    sim.close()
    return first, before, reason, big


@pytest.mark.parametrize("cfg", ["cfg1", "cfg2", "cfg3", "cfg4", "cupboard", "cfg3+solo", "cupboard+solo"])
def test_env_step_300_matches_oracle(models, cfg):
    """(The "+solo" cases: the same with every env handed over to a solo server after ten substeps - the migration forced on every env.)
    A whole env-step (300 substeps): median |dobs| < 1e-4, 90th percentile < 2e-3 (contact dynamics amplify fp32 rounding over
    300 steps; per-substep parity is the sharp test; in the cupboard scene a random ctrl drives the arm into the doors in a fifth of
    the envs, which is chaotic in either precision: there the percentile bound is on the 75th).  And no env beyond 2e-3 goes
    unexplained: both sides are replayed substep by substep and the env must show a substep at which the HIP contact LIST differs from
    the oracle's beyond the stage tolerances - a contact that exists in one precision only, or (flat finger face on a flat door: the
    single MPR point of a face-face pair is not unique) one whose position / normal / depth the two precisions place differently; from there
    on the two runs are different trajectories - with the two still within 2e-3 of each other up to that substep."""
    solo = cfg.endswith("+solo")          # every env handed over to a solo server after the first round of the work queue (round 4)
    cfg = cfg.split("+")[0]
    m = models[cfg]
    n = 64
    rng = np.random.default_rng(12)
    q, v, ctrl = random_states(m, n, rng)
    sim = hs.BatchSim(m, n)
    if solo:
        assert sim.set_solo(n, 0.01)
        sim.set_queue(1, 10)
    sim.set_state(np.zeros(n), q, np.zeros_like(v))
    obs, rew, done, ns = sim.step(ctrl, 300)
    if solo:
        assert sim.solo_handovers() == n
    errs = []
    for e in range(n):
        o = OracleSim(m)
        o.qpos[:] = q[e]
        o.env_step(ctrl[e], 300)
        errs.append(np.abs(obs[e] - np.concatenate([o.qpos, o.qvel])).max())
    errs = np.array(errs)
    assert np.median(errs) < 1e-4, errs
    assert np.percentile(errs, 75 if cfg == "cupboard" else 90) < 2e-3, errs
    assert (ns == 300).all()
    sim.close()
    out = np.flatnonzero(errs >= 2e-3)
    if solo:
        # the substep-by-substep replay below runs in the plain mode and cannot retrace a solo run (the replicas sum the Hessian in another order,
        # and an env under a kN contact amplifies that): the solo case is held to the percentile bounds above and to as many envs beyond 2e-3 as
        # the plain run of the same inputs has, plus two
        plain = hs.BatchSim(m, n)
        plain.set_state(np.zeros(n), q, np.zeros_like(v))
        obs_p = plain.step(ctrl, 300)[0]
        plain.close()
        errs_p = np.array([np.abs(obs_p[e] - obs[e]).max() for e in range(n)])
        n_plain = 0
        for e in range(n):
            o = OracleSim(m); o.qpos[:] = q[e]; o.env_step(ctrl[e], 300)
            n_plain += np.abs(obs_p[e] - np.concatenate([o.qpos, o.qvel])).max() >= 2e-3
        print(f"{cfg}+solo: {len(out)} envs beyond 2e-3 of the oracle ({int(n_plain)} in the plain run); solo vs plain run: median {np.median(errs_p):.1e}, max {errs_p.max():.1e}")
        assert len(out) <= n_plain + 2
        return
    if len(out):
        first, before, reason, big = first_contact_divergence(m, q[out], ctrl[out], 300)
        print(f"{cfg}: {len(out)} of {n} envs beyond 2e-3 after 300 substeps; first substep with a differing contact list {first.tolist()}, "
              f"largest |dobs| before it {np.array2string(before, precision=1)}, what differed: {reason}; first (substep, obs index, |dobs|, HIP Newton iterations, oracle's) beyond 2e-3 while the lists still agreed: {big}")
        assert (first >= 0).all(), f"envs {out[first < 0].tolist()} are beyond 2e-3 without any divergence of the contact sets"
        assert (before < 2e-3).all(), (out.tolist(), before)
    else:
        print(f"{cfg}: no env beyond 2e-3 after 300 substeps")


def test_goal_early_exit_matches_oracle(models):
    """hsr/env.py:124-131: per-env done latch at the first substep whose block xpos is inside the
    geofence; finished envs stop integrating; reward = float(done).  The latch (done, reward, substep count within one)
    must agree for every env; the returned state agrees to |dobs| < 1e-3 for >= 95 % of the envs that stopped at the same
    substep and to 1e-2 for all of them (a quarter of the blocks is dropped from 5 cm and bounces for the 120 substeps)."""
    m = models["cfg2"]
    n = 64
    rng = np.random.default_rng(13)
    q, v, ctrl = random_states(m, n, rng)
    bid = m.body_id(m.block_body())
    # goals: half of them right at the block (immediate success), half away; some blocks dropped from above
    goal = q[:, 2:5].copy()
    goal[n // 2:, 0] += 0.3
    q[::4, 4] += 0.05                      # falls ~0.05 m: enters a 0.03 geofence after some substeps
    goal[::4, 2] = 0.422
    sim = hs.BatchSim(m, n)
    sim.reset(qpos0=q, mocap=goal)
    obs, rew, done, ns = sim.step(ctrl, 120, bid, 0.03)
    errs = []
    for e in range(n):
        o = OracleSim(m)
        o.qpos[:] = q[e]; o.mocap_pos[:] = goal[e]
        k, dn = o.env_step(ctrl[e], 120, bid, goal[e], 0.03)
        assert dn == bool(done[e]) and rew[e] == float(dn), e
        assert abs(k - ns[e]) <= (1 if dn else 0), (e, k, ns[e])
        if k == ns[e]:
            errs.append(np.abs(obs[e] - np.concatenate([o.qpos, o.qvel])).max())
    errs = np.array(errs)
    assert (errs < 1e-3).mean() >= 0.95 and errs.max() < 1e-2, np.sort(errs)[-5:]
    assert done.any() and not done.all() and (ns[done] < 120).any()
    sim.close()


def test_masked_reset_and_body_xpos(models):
    m = models["cfg3"]
    n = 32
    sim = hs.BatchSim(m, n)
    rng = np.random.default_rng(14)
    q, v, ctrl = random_states(m, n, rng)
    sim.step(ctrl, 5)
    t0, q0, v0 = sim.get_state()
    mask = np.zeros(n, np.uint8); mask[::2] = 1
    sim.reset(mask=mask)
    t1, q1, v1 = sim.get_state()
    assert np.allclose(q1[::2], m.qpos0, atol=1e-6) and np.allclose(v1[::2], 0) and np.allclose(t1[::2], 0)
    assert np.array_equal(q1[1::2], q0[1::2]) and np.array_equal(v1[1::2], v0[1::2])
    o = OracleSim(m)
    o.forward()
    for name in ("block0", "hand_l_distal_link", "hand_r_distal_link", "goal"):
        got = sim.body_xpos(m.body_id(name))[0]
        assert np.allclose(got, o.body_xpos(m.body_id(name)), atol=2e-6), name
    sim.close()


def test_determinism(models):
    """Same inputs -> bit-identical outputs across runs of the persistent kernel (hipGraph replay of the per-substep chain is
    compared with plain launches in test_gpu_hotpath.py)."""
    m = models["cfg3"]
    n = 128
    rng = np.random.default_rng(15)
    q, v, ctrl = random_states(m, n, rng)
    outs = []
    for rep in range(3):
        sim = hs.BatchSim(m, n)
        assert sim.is_persistent()
        sim.set_state(np.zeros(n), q, v)
        o1 = sim.step(ctrl, 50)[0]
        o2 = sim.step(ctrl, 50)[0]
        outs.append((o1.copy(), o2.copy()))
        sim.close()
    for k in (1, 2):
        for a, b in zip(outs[0], outs[k]):
            assert np.array_equal(a, b)


def test_full_size_invariants(models):
    """BASELINE size (8192 envs, cfg3): size-independent properties - unit quaternions, finite state,
    blocks stay on/above the pan, done == (distance < geofence) at the reported state."""
    m = models["cfg3"]
    n = 8192
    rng = np.random.default_rng(16)
    q, v, ctrl = random_states(m, n, rng)
    goal = np.column_stack([rng.uniform(-0.1, 0.1, n), rng.uniform(-0.2, 0.2, n), np.full(n, 0.422)])
    sim = hs.BatchSim(m, n)
    sim.reset(qpos0=q, mocap=goal)
    obs, rew, done, ns = sim.step(ctrl, 300, m.body_id(m.block_body()), 0.05)
    assert np.isfinite(obs).all() and not sim.bad_state()[1]
    assert np.abs(np.linalg.norm(obs[:, 10:14], axis=1) - 1).max() < 1e-5
    assert (obs[:, 9] > 0.40).mean() > 0.99
    assert ((ns == 300) | done).all() and (rew == done.astype(np.float32)).all()
    # replicas: the same env state in different batch slots gives identical results
    sim2 = hs.BatchSim(m, 256)
    sim2.reset(qpos0=q[:256], mocap=goal[:256])
    obs2 = sim2.step(ctrl[:256], 300, m.body_id(m.block_body()), 0.05)[0]
    assert np.array_equal(obs2, obs[:256])
    sim.close(); sim2.close()


@pytest.mark.parametrize("cfg", ["cfg2", "cfg3", "cfg4", "cupboard"])
def test_persistent_kernel_matches_per_substep_kernels(models, cfg):
    """The whole-env-step persistent kernel (k_env_step_mf) and the per-substep kernel chain (k_kinematics, k_cull,
    k_narrow, k_solve_mf) restate the same substep; they share the solver body and the narrowphase routines but not the
    kinematics / cull code, so results are close but not bit-identical and contact-rich envs amplify the difference.
    fp32 tolerance after 60 substeps: |dqpos| < 2e-5, |dqvel| < 2e-3 for >= 95 % of envs (median |dqpos| < 1e-6), and
    the goal latch (done, nsteps) agrees wherever the states agree."""
    m = models[cfg]
    n = 256
    rng = np.random.default_rng(21)
    q, v, ctrl = random_states(m, n, rng)
    goal = np.column_stack([rng.uniform(-0.1, 0.1, n), rng.uniform(-0.2, 0.2, n), np.full(n, 0.422)])
    res = []
    for persistent in (True, False):
        sim = hs.BatchSim(m, n)
        if persistent and not sim.is_persistent():
            sim.close()
            pytest.skip("model does not fit the persistent kernel")
        sim.set_persistent(persistent)
        assert sim.is_persistent() == persistent
        sim.reset(qpos0=q, mocap=goal)
        obs, rew, done, ns = sim.step(ctrl, 60, m.body_id(m.block_body()), 0.02)
        res.append((obs.copy(), done.copy(), ns.copy()))
        assert not sim.bad_state()[1]
        sim.close()
    (oa, da, na), (ob, db, nb) = res
    same_len = na == nb
    assert same_len.mean() >= 0.98
    dq = np.abs(oa[:, :m.nq] - ob[:, :m.nq]).max(axis=1)
    dv = np.abs(oa[:, m.nq:] - ob[:, m.nq:]).max(axis=1)
    ok = (dq < 2e-5) & (dv < 2e-3)
    assert ok[same_len].mean() >= 0.95, (np.sort(dq)[-5:], np.sort(dv)[-5:])
    assert np.median(dq) < 1e-6
    assert (da == db)[same_len & ok].all()


@pytest.mark.parametrize("cfg", ["cfg3", "cupboard"])
def test_openai_observation_matches_numpy_restatement(models, cfg):
    """Fused obs kernel (hsr_batch_obs_openai) against the fp64 numpy restatement in tests/oracle_batch.py:
    (1) right after set_state + forward, |d| < 2e-5; (2) after an env-step of K substeps the body poses / velocities are
    those of the K-th forward pass and the joint values the integrated ones - compared with the oracle stepped K-1 and
    K times, fp32 tolerance 2e-4 for >= 95 % of the envs."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).parent))
    from oracle_batch import openai_obs_reference
    m = models[cfg]
    n = 48
    rng = np.random.default_rng(31)
    q, v, ctrl = random_states(m, n, rng)
    sim = hs.BatchSim(m, n)
    sim.set_state(np.zeros(n), q, v)
    sim.forward()
    got = sim.obs_openai()
    ref = np.array([openai_obs_reference(m, q[e], v[e], q[e], v[e]) for e in range(n)])
    assert got.shape == (n, 25) and np.abs(got - ref).max() < 2e-5, np.abs(got - ref).max(axis=0)
    K = 12
    sim.step(ctrl, K)
    got = sim.obs_openai()
    ref = []
    for e in range(n):
        o = OracleSim(m)
        o.qpos[:] = q[e]; o.qvel[:] = v[e]; o.ctrl[:] = ctrl[e]
        for _ in range(K - 1):
            o.step()
        qf, vf = o.qpos.copy(), o.qvel.copy()
        o.step()
        ref.append(openai_obs_reference(m, qf, vf, o.qpos, o.qvel))
    err = np.abs(got - np.array(ref)).max(axis=1)
    assert (err < 2e-4).mean() >= 0.95, np.sort(err)[-6:]
    sim.close()


@pytest.mark.parametrize("cfg", ["cfg3", "cfg4"])
def test_ragged_batch_sizes(models, cfg):
    """Batch sizes that do not fill the last workgroup (4 envs per wave at cfg3, 2 at cfg4) and the single-env case:
    every env of a ragged batch gives bit-identical results to the same env inside a full batch."""
    m = models[cfg]
    rng = np.random.default_rng(41)
    q, v, ctrl = random_states(m, 68, rng)
    goal = np.column_stack([rng.uniform(-0.1, 0.1, 68), rng.uniform(-0.2, 0.2, 68), np.full(68, 0.422)])
    bid = m.body_id(m.block_body())
    full = hs.BatchSim(m, 68)
    full.reset(qpos0=q, mocap=goal)
    ref = full.step(ctrl, 40, bid, 0.03)
    full.close()
    for n in (1, 5, 67):
        sim = hs.BatchSim(m, n)
        sim.reset(qpos0=q[:n], mocap=goal[:n])
        got = sim.step(ctrl[:n], 40, bid, 0.03)
        for a, b in zip(got, ref):
            assert np.array_equal(a, b[:n]), n
        assert not sim.bad_state()[1]
        sim.close()
