"""What the solver is FED, checked without the oracle's own assembly code (CPU).

tests/test_oracle_optimality.py checks the oracle's SOLVER against an independent minimiser, but copies the problem (J, aref, R, mu) out of the
oracle's forward pass.  Here the constraint assembly behind `self.sim.step()` (hsr/env.py:123: mj_makeConstraint, mj_referenceConstraint) is
restated in numpy from the oracle's KINEMATIC outputs only - link poses `xpos / xmat`, the contact list (position, frame, distance, geoms) and
`qvel` - and from the model tables the compiler wrote (pair solref / solimp / friction / condim, inverse weights):
  * Jacobian rows by central differences: every dof is displaced by +-eps (free-joint rotations through the quaternion, about the body axes, as
    mju_quatIntegrate does), the kinematics are re-run, and a row is the frame axis dotted with the difference of the two links' contact-point
    displacements (rows 0-2) or of their rotations (rows 3-5) - nothing of `point_jac_row` / `dof_point_vel` is used;
  * reference acceleration `aref = -B (J qvel) - K imp dist` with `B = 2 / (dmax tc)`, `K = 1 / (dmax^2 tc^2 dr^2)` and MuJoCo's published
    impedance sigmoid restated below; regularisers `R0 = (1 - imp) / imp * (invweight of the two geoms)`, friction rows `R0 / impratio` and
    `R1 f0^2 / f_j^2`, `mu = f0 sqrt(R1 / R0)`; limit rows `J = +-e_dof`, `R = (1 - imp) / imp * dof_invweight0`.
on the 250 hard states of tests/golden/solver_states.npz.  Measured (this container, `-s` prints it): |J - J_fd| <= 4.3e-10, R and mu equal to the
last bit, aref 1.5e-7 relative (the finite-difference error of J times B ~ 200) over 7 015 rows."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from hsr_env_amd.compiler import load_config          # noqa: E402
from oracle.oracle import OracleSim                    # noqa: E402

FIX = ROOT / "tests" / "golden" / "solver_states.npz"
MINVAL, MINIMP, MAXIMP = 1e-15, 1e-4, 0.9999


def impedance(solimp, pos):
    """MuJoCo 2.0 `solimp = (d0, dwidth-end value dmax, width, midpoint, power)`: impedance d(|pos| / width) on a sigmoid from dmin to dmax."""
    dmin, dmax, width, mid, power = [float(x) for x in solimp]
    dmin, dmax = np.clip(dmin, MINIMP, MAXIMP), np.clip(dmax, MINIMP, MAXIMP)
    width, mid, power = max(width, MINVAL), np.clip(mid, MINIMP, MAXIMP), max(power, 1.0)
    if dmin == dmax or width <= MINVAL:
        return 0.5 * (dmin + dmax)
    x = abs(pos) / width
    if x >= 1:
        return dmax
    if x <= 0:
        return dmin
    if power == 1:
        y = x
    elif x <= mid:
        y = x ** power / mid ** (power - 1)
    else:
        y = 1 - (1 - x) ** power / (1 - mid) ** (power - 1)
    return dmin + y * (dmax - dmin)


def displaced(m, qpos, k, eps):
    """qpos after moving dof k by eps (mj_integratePos for one coordinate)."""
    q = qpos.copy()
    l = int(m.dof_link[k]); j = k - int(m.link_dofadr[l])
    if not m.link_free[l]:
        q[int(m.dof_qposadr[k])] += eps
        return q
    qa = int(m.link_qposadr[l])
    if j < 3:
        q[qa + j] += eps
        return q
    w, x, y, z = q[qa + 3:qa + 7]
    d = np.zeros(4); d[0] = np.cos(eps / 2); d[1 + (j - 3)] = np.sin(eps / 2)          # rotation by eps about body axis j - 3
    q[qa + 3:qa + 7] = [w * d[0] - x * d[1] - y * d[2] - z * d[3], w * d[1] + x * d[0] + y * d[3] - z * d[2],
                        w * d[2] - x * d[3] + y * d[0] + z * d[1], w * d[3] + x * d[2] - y * d[1] + z * d[0]]
    return q


def poses(o, qpos):
    o.qpos[:] = qpos
    o.forward()
    return o.xpos.copy(), o.xmat.copy()


def test_constraint_rows_follow_from_the_kinematics_alone():
    st = np.load(FIX)
    sims = {}
    worst = dict(J=0.0, aref=0.0, R=0.0, mu=0.0)
    nrows = 0
    for i in range(len(st["cfg"])):
        cfg = str(st["cfg_names"][st["cfg"][i]])
        if cfg not in sims:
            sims[cfg] = (load_config(cfg), OracleSim(load_config(cfg)), OracleSim(load_config(cfg)))
        m, o, o2 = sims[cfg]
        nq, nv = m.nq, m.nv
        qpos, qvel = st["qpos"][i, :nq].copy(), st["qvel"][i, :nv].copy()
        o.qpos[:] = qpos; o.qvel[:] = qvel; o.ctrl[:] = st["ctrl"][i, :m.nu]; o.qacc_warmstart[:] = st["warm"][i, :nv]
        o.forward()
        efc, cons, nlim = o.efc(), o.contacts(), o.solver_stats()[3]
        xpos0, xmat0 = o.xpos.copy(), o.xmat.copy()
        qn = o.qpos.copy()                                       # (the forward pass normalised the quaternions)
        opt = m.arrays["opt"]
        impratio = float(opt[1])
        # --- per-dof displacement fields of every link by central differences of the kinematics
        eps = 1e-6
        dpos, drot = np.zeros((nv, m.nlink, 3)), np.zeros((nv, m.nlink, 3))          # d x_l / dq_k, and the rotation vector of link l per unit of q_k
        Rp, Rm = np.zeros((nv, m.nlink, 3, 3)), np.zeros((nv, m.nlink, 3, 3))
        xp, xm = np.zeros((nv, m.nlink, 3)), np.zeros((nv, m.nlink, 3))
        for k in range(nv):
            xp[k], Rp[k] = poses(o2, displaced(m, qn, k, eps))
            xm[k], Rm[k] = poses(o2, displaced(m, qn, k, -eps))
            for l in range(m.nlink):
                dR = Rp[k, l] @ Rm[k, l].T
                drot[k, l] = np.array([dR[2, 1] - dR[1, 2], dR[0, 2] - dR[2, 0], dR[1, 0] - dR[0, 1]]) / (4 * eps)
        J = np.zeros((o.nefc, nv)); aref = np.zeros(o.nefc); R = np.zeros(o.nefc)
        # --- limit rows, in dof order, lower side then upper side
        r = 0
        for k in range(nv):
            if not m.dof_limited[k]:
                continue
            q = qn[int(m.dof_qposadr[k])]
            for side, dist in ((0, q - m.dof_range[k][0]), (1, m.dof_range[k][1] - q)):
                if dist < 0:
                    sg = 1.0 if side == 0 else -1.0
                    J[r, k] = sg
                    si, (tc, dr) = m.dof_solimp[k], m.dof_solref[k]
                    imp, dmax = impedance(si, dist), np.clip(si[1], MINIMP, MAXIMP)
                    aref[r] = -2.0 / (dmax * tc) * sg * qvel[k] - imp / (dmax * dmax * tc * tc * dr * dr) * dist
                    R[r] = max((1 - imp) / imp * m.dof_invweight0[k], MINVAL)
                    r += 1
        assert r == nlim, (i, r, nlim)
        # --- contact rows
        pairs = {(int(a), int(b)): p for p, (a, b) in enumerate(zip(m.pair_geom1, m.pair_geom2))}
        for ci, c in enumerate(cons):
            pos, frame, dist, g1, g2, dim, mu_o = c[:3], c[3:12].reshape(3, 3), c[12], int(c[13]), int(c[14]), int(c[15]), c[16]
            p = pairs[(g1, g2)]
            assert dim == int(m.pair_condim[p])
            l1, l2 = int(m.geom_link[g1]), int(m.geom_link[g2])
            fri, si, (tc, dr) = m.pair_friction[p], m.pair_solimp[p], m.pair_solref[p]
            imp, dmax = impedance(si, dist), np.clip(si[1], MINIMP, MAXIMP)
            tran = m.geom_invweight[g1][0] + m.geom_invweight[g2][0]
            for j in range(dim):
                ax = frame[j % 3]
                for k in range(nv):
                    if j < 3:
                        # the contact point as a material point of each link, at q + eps and q - eps
                        d = []
                        for l in (l1, l2):
                            loc = xmat0[l].T @ (pos - xpos0[l])
                            d.append(((xp[k, l] + Rp[k, l] @ loc) - (xm[k, l] + Rm[k, l] @ loc)) / (2 * eps))
                        J[r + j, k] = ax @ (d[1] - d[0])
                    else:
                        J[r + j, k] = ax @ (drot[k, l2] - drot[k, l1])
            R0 = max((1 - imp) / imp * tran, MINVAL)
            R[r] = R0
            mu = fri[0]
            if dim > 1:
                R[r + 1] = R0 / max(impratio, MINVAL)
                for j in range(2, dim):
                    R[r + j] = R[r + 1] * fri[0] ** 2 / fri[j - 1] ** 2
                mu = fri[0] * np.sqrt(R[r + 1] / R0)
            B, K = 2.0 / (dmax * tc), 1.0 / (dmax * dmax * tc * tc * dr * dr)
            jv = J[r:r + dim] @ qvel
            aref[r:r + dim] = -B * jv
            aref[r] -= K * imp * dist
            worst["mu"] = max(worst["mu"], abs(mu - mu_o) / mu)
            r += dim
        assert r == o.nefc, (i, r, o.nefc)
        nrows += r
        worst["J"] = max(worst["J"], float(np.abs(J - efc["J"]).max()) if r else 0.0)
        worst["aref"] = max(worst["aref"], float((np.abs(aref - efc["aref"]) / (1 + np.abs(efc["aref"]))).max()) if r else 0.0)
        worst["R"] = max(worst["R"], float((np.abs(R - efc["R"]) / efc["R"]).max()) if r else 0.0)
    print("rows", nrows, worst)
    assert nrows > 4000
    # (aref carries the finite-difference error of J times B = 2 / (dmax tc) ~ 200 times |qvel|)
    assert worst["J"] < 2e-8 and worst["aref"] < 1e-6 and worst["R"] < 1e-12 and worst["mu"] < 1e-12, worst
