"""Known-answer tests pinning the CPU oracle (the reference ships no tests and
MuJoCo is absent, so these analytic cases are the only pins of the restated algorithm)."""
import numpy as np
import pytest

from hsr_env_amd import compiler as hc
from oracle.oracle import OracleSim

H = 0.002
G = 9.81


def test_free_fall_closed_form(models):
    """KAT 1: block in free flight, semi-implicit Euler: v_n = -g h n, z_n = z0 - g h^2 n(n+1)/2."""
    m = models["cfg2"]
    s = OracleSim(m)
    s.qpos[2:5] = [0.0, 0.0, 1.5]          # well above the pan, nothing near
    n = 100
    for _ in range(n):
        s.step()
    assert s.ncon == 0
    assert abs(s.qvel[4] - (-G * H * n)) < 1e-12
    assert abs(s.qpos[4] - (1.5 - G * H * H * n * (n + 1) / 2)) < 1e-12
    assert np.allclose(s.qpos[[2, 3]], 0) and np.allclose(s.qpos[5:9], [1, 0, 0, 0])


def test_slide_x_implicit_damping_recurrence(models):
    """KAT 2: slide_x alone (damping 2200, kp 300, gear 3, force clamp +-100; world.xml:104-106,
    hsr.mjcf:4): (M + h B) a = gear*clamp(kp*(ctrl - gear q)) - B v ; v += h a ; q += h v."""
    m = models["cfg1"]
    s = OracleSim(m)
    M = hc.mass_matrix(m, m.qpos0)[0, 0]
    B, kp, gear = 2200.0, 300.0, 3.0
    ctrl = 0.3
    s.ctrl[:] = [ctrl, 0.0]
    q = v = 0.0
    for _ in range(500):
        s.step()
        f = np.clip(kp * (ctrl - gear * q), -100, 100)
        a = (gear * f - B * v) / (M + H * B)
        v += H * a
        q += H * v
        assert abs(s.qpos[0] - q) < 1e-12 and abs(s.qvel[0] - v) < 1e-12
    # steady state q* = ctrl / gear (inside range -.12 .. .22)
    for _ in range(20000):
        s.step()
    assert abs(s.qpos[0] - ctrl / gear) < 1e-6


def test_slide_range_limits_hold(models):
    """Steady state is limited by the joint ranges slide_x >= -.12, slide_y >= -.22 (hsr.mjcf:4,6):
    ctrl=-1 asks for q=-1/3.  (Towards +x the base hull meets the pan edge first, see next test.)"""
    m = models["cfg1"]
    s = OracleSim(m)
    s.ctrl[:] = [-1.0, -1.0]
    for _ in range(20000):
        s.step()
    assert -0.1202 < s.qpos[0] < -0.12 and -0.2202 < s.qpos[1] < -0.22   # soft limit: small violation
    assert np.allclose(s.qvel, 0, atol=1e-8)


def test_base_hull_stops_at_pan_edge(models):
    """Driving +x, the convex hull of base.stl meets the pan edge (x=-.17, z=.395) before the
    joint limit .22: one MPR contact pan->base, robot held near slide_x = .16."""
    m = models["cfg1"]
    s = OracleSim(m)
    s.ctrl[:] = [1.0, 0.0]
    for _ in range(20000):
        s.step()
    assert s.ncon == 1
    c = s.contacts()[0]
    assert m.names["geom"][int(c[14])] == "base_link:base" and int(c[13]) == 1
    assert abs(c[0] - (-0.17)) < 1e-3 and c[3] < -0.8
    assert 0.15 < s.qpos[0] < 0.17 and abs(s.qvel[0]) < 1e-6


def test_block_rests_on_pan(models):
    """KAT 3: block at rest on the pan: z -> 0.422 - delta, delta = m g / (4 * D * K * imp) with
    D = imp/((1-imp)/m), K = 1/(dmax^2 tc^2): no drift in x, y, yaw over 300 substeps."""
    m = models["cfg2"]
    s = OracleSim(m)
    for _ in range(3000):
        s.step()
    imp, tc = 0.99, 0.01
    D = 1.0 / ((1 - imp) / imp * 1.0)
    K = 1.0 / (imp * imp * tc * tc)
    delta = 1.0 * G / 4 / (D * K * imp)
    assert s.ncon == 4
    assert abs(s.qpos[4] - (0.422 - delta)) < 1e-8
    assert np.allclose(s.qpos[[2, 3]], 0, atol=1e-10)
    assert np.allclose(s.qpos[5:9], [1, 0, 0, 0], atol=1e-10)
    assert np.allclose(s.qvel, 0, atol=1e-7)      # solver tolerance 1e-8 leaves a tiny residual


def test_block_rests_rotated(models):
    """Same with a yawed, offset block: still 4 contacts, no drift."""
    m = models["cfg2"]
    s = OracleSim(m)
    yaw = 0.7
    s.qpos[2:9] = [0.05, -0.1, 0.422, np.cos(yaw / 2), 0, 0, np.sin(yaw / 2)]
    q0 = s.qpos.copy()
    for _ in range(600):
        s.step()
    assert s.ncon == 4
    assert np.allclose(s.qpos[[2, 3]], q0[[2, 3]], atol=1e-8)
    assert np.allclose(s.qpos[5:9], q0[5:9], atol=1e-8)
    assert abs(s.qpos[4] - 0.422) < 1e-5


def test_arm_lift_ctrl_clamped(models):
    """KAT 4: action 0 on arm_lift is clamped to ctrlrange lo 2.3 (world.xml:112)."""
    m = models["cfg3"]
    s = OracleSim(m)
    s.forward()
    # actuator force = clamp(kp*2.3 - kp*gear*q, +-20) = 20 -> qfrc = gear*20 = 100 on dof 2
    a = OracleSim(m)
    a.ctrl[2] = 2.3
    a.forward()
    assert np.allclose(s.qacc_smooth, a.qacc_smooth)
    b = OracleSim(m)
    b.ctrl[2] = 100.0      # clamped to 4.1 -> same saturated force
    b.forward()
    assert np.allclose(s.qacc_smooth, b.qacc_smooth)


def test_finger_limit_hold(models):
    """KAT 5: hand_l_proximal driven past its 0..0.349066 rad range stays at the soft limit."""
    m = models["cfg3"]
    s = OracleSim(m)
    s.ctrl[:] = [0, 0, 2.3, -0.5, 0, 5.0, -5.0]   # clamped to 0.349066 / 0
    for _ in range(3000):
        s.step()
    assert abs(s.qpos[5] - 0.349066) < 5e-3
    assert abs(s.qpos[6] - 0.0) < 5e-3
    assert not s.bad


def test_quaternion_unit_norm_and_determinism(models):
    m = models["cfg3"]
    rng = np.random.default_rng(0)
    runs = []
    for rep in range(2):
        s = OracleSim(m)
        s.qpos[7:10] = [0.02, 0.05, 0.6]
        s.qvel[10:13] = [3.0, -2.0, 5.0]
        rng = np.random.default_rng(0)
        for i in range(400):
            if i % 100 == 0:
                s.ctrl[:] = rng.uniform(m.act_ctrlrange[:, 0], m.act_ctrlrange[:, 1])
            s.step()
            assert abs(np.linalg.norm(s.qpos[10:14]) - 1) < 1e-12
        runs.append((s.qpos.copy(), s.qvel.copy()))
        assert not s.bad
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1])


def test_free_spin_conserves_angular_momentum(models):
    """Torque-free tumbling block: world-frame angular momentum R I w is conserved (gyroscopic bias)."""
    m = models["cfg2"]
    s = OracleSim(m)
    s.qpos[2:5] = [0, 0, 50.0]
    s.qvel[5:8] = [4.0, 1.0, -3.0]
    I = m.link_inertia[2, :3]

    def L():
        R = hc.quat_to_mat(s.qpos[5:9])
        return R @ (I * s.qvel[5:8])
    L0 = L()
    for _ in range(500):
        s.step()
    assert np.allclose(L(), L0, rtol=2e-2)     # first-order integrator: <1 % drift in 1 s


def test_bias_matches_lagrangian(models):
    """qfrc_bias on the robot dofs equals C(q,v)v + dV/dq from finite differences of the numpy
    mass matrix / potential energy (independent of the oracle's RNE)."""
    m = models["cfg3"]
    nr = 7
    rng = np.random.default_rng(1)
    q = m.qpos0.copy()
    q[:nr] = [0.05, -0.03, 0.1, -0.8, 0.4, 0.2, 0.1]
    v = np.zeros(m.nv); v[:nr] = rng.normal(size=nr) * [0.2, 0.2, 0.2, 1.5, 2.0, 2.0, 2.0]

    def Mq(qq): return hc.mass_matrix(m, qq)[:nr, :nr]

    def V(qq):
        xpos, xquat = hc.link_kinematics(m, qq)
        e = 0.0
        for l in range(1, m.nlink - 1):
            c = xpos[l] + hc.quat_to_mat(xquat[l]) @ m.link_com[l]
            e += m.link_mass[l] * G * c[2]
        return e
    eps = 1e-6
    dM = np.zeros((nr, nr, nr)); dV = np.zeros(nr)
    for k in range(nr):
        qp, qm = q.copy(), q.copy(); qp[k] += eps; qm[k] -= eps
        dM[:, :, k] = (Mq(qp) - Mq(qm)) / (2 * eps)
        dV[k] = (V(qp) - V(qm)) / (2 * eps)
    vr = v[:nr]
    C = np.einsum("ijk,j,k->i", dM, vr, vr) - 0.5 * np.einsum("jki,j,k->i", dM, vr, vr)
    s = OracleSim(m)
    s.qpos[:] = q; s.qvel[:] = v
    s.forward()
    assert np.allclose(s.qfrc_bias[:nr], C + dV, rtol=1e-5, atol=1e-5)


def test_mass_matrix_matches_numpy(models):
    for name, m in models.items():
        s = OracleSim(m)
        rng = np.random.default_rng(2)
        q = m.qpos0.copy()
        nrob = m.nu
        q[:nrob] = rng.uniform(-0.2, 0.2, nrob)
        s.qpos[:] = q
        s.forward()
        assert np.allclose(s.M, hc.mass_matrix(m, s.qpos), atol=1e-10), name


def test_tilted_block_settles_flat(models):
    """Dropped tilted block comes to rest flat on the pan (exercises box-box edge/face cases)."""
    m = models["cfg2"]
    s = OracleSim(m)
    ang = 0.4
    s.qpos[2:9] = [0.0, 0.0, 0.50, np.cos(ang / 2), np.sin(ang / 2), 0, 0]
    for _ in range(2500):
        s.step()
    assert not s.bad
    assert abs(s.qpos[4] - 0.422) < 2e-4 or abs(s.qpos[4] - 0.43) < 2e-3 or abs(s.qpos[4] - 0.455) < 2e-3
    assert np.linalg.norm(s.qvel[2:8]) < 1e-3


def test_env_step_early_exit(models):
    """hsr/env.py:118-131: goal test after every substep, break on success."""
    m = models["cfg2"]
    s = OracleSim(m)
    bid = m.body_id("block0")
    n, done = s.env_step(np.zeros(2), 300, bid, np.array([0, 0, 0.422]), 0.05)
    assert done and n == 1
    s = OracleSim(m)
    n, done = s.env_step(np.zeros(2), 300, bid, np.array([0.3, 0, 0.422]), 0.05)
    assert (not done) and n == 300
    n, done = s.env_step(np.zeros(2), 7, -1, None, 0.0)
    assert (not done) and n == 7
