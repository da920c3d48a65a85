"""GPU twins of tests/test_narrowphase_geometry.py: the HIP path's box-box and MPR contacts, read through the
C-ABI, against the brute-force SAT and the exact penetration depth of tests/geom_checks.py - for the persistent kernel's own narrowphase (the
8-lane box-box, the 8-lane MPR sub-groups: `step(ctrl, 1)` with the introspection flag) and for the per-substep chain (`forward()`).
Behind `self.sim.step()` (hsr/env.py:123 -> mj_collision)."""
import numpy as np
import pytest

from hsr_env_amd import sim as hs
from oracle.oracle import OracleSim
import geom_checks as gc
from test_narrowphase_geometry import boxbox_states, check_boxbox, check_mpr_bounds, mpr_compare, mpr_samples, report

pytestmark = pytest.mark.gpu


def pair_rows(m, con_e, g1, g2):
    """the used contact slots of pair (g1, g2) of one env: rows (pos3, normal3, dist)"""
    p = [k for k in range(m.npair) if int(m.pair_geom1[k]) == g1 and int(m.pair_geom2[k]) == g2][0]
    rows = con_e[int(m.pair_slot[p]):int(m.pair_slot[p + 1])]
    return rows[rows[:, 6] <= 0].astype(np.float64)


def hip_contacts(m, q, persistent):
    n = q.shape[0]
    sim = hs.BatchSim(m, n)
    sim.set_state(np.zeros(n), q, np.zeros((n, m.nv)))
    if persistent:
        assert sim.is_persistent()
        sim.set_debug(True)
        sim.step(np.tile(m.act_ctrlrange.mean(1), (n, 1)), 1)
    else:
        sim.forward()
    con = sim.get_field(hs.F_CONTACT)
    assert not sim.bad_state()[1]
    sim.close()
    return con


@pytest.mark.parametrize("persistent", [True, False])
def test_box_box_against_brute_force_sat(models, persistent):
    m = models["cfg4"]
    rng = np.random.default_rng(5)
    n = 3072
    q = boxbox_states(m, n, rng).astype(np.float32).astype(np.float64)          # what the device holds
    con = hip_contacts(m, q, persistent)
    o = OracleSim(m)
    stats = dict(separated=0, face=0, edge=0, touching_without_points=0, overlap_without_points=0, deepest_vertex_reported=0, edge_point_outside=0)
    for e in range(n):
        o.qpos[:] = q[e]; o.qvel[:] = 0
        o.forward()                                  # (only its kinematics are used: where the boxes are)
        for g1, g2 in ((17, 18), (1, 19)):
            check_boxbox(m, o.xpos, o.xmat.reshape(-1, 3, 3), pair_rows(m, con[e], g1, g2), g1, g2, 1e-6, stats)
    print(stats)
    assert stats["face"] > 2500 and stats["edge"] > 250 and stats["deepest_vertex_reported"] > 1200
    assert stats["overlap_without_points"] <= 0.01 * 2 * n, stats


@pytest.mark.parametrize("persistent", [True, False])
def test_mpr_depth_against_the_exact_penetration_depth(models, persistent):
    m = models["cfg3"]
    rng = np.random.default_rng(9)
    samples, xpos, xmat = mpr_samples(m, rng, 40)
    q = np.array([s[1] for s in samples]).astype(np.float32).astype(np.float64)
    con = hip_contacts(m, q, persistent)
    pairs = {(int(a), int(b)) for a, b in zip(m.pair_geom1, m.pair_geom2)}
    rel, absd, ang, ff, missed, n_pen = [], [], [], [], 0, 0
    for e, (g, _, face_first) in enumerate(samples):
        g1, g2 = (17, g) if (17, g) in pairs else (g, 17)
        rows = pair_rows(m, con[e], g1, g2)
        de, dm, a = mpr_compare(m, xpos, xmat, g, q[e], rows[0] if len(rows) else None)
        if de <= 2e-6:
            continue
        n_pen += 1
        if not len(rows):
            missed += 1
            continue
        assert dm >= de - 5e-6, ("MPR below the exact depth", g, de, dm)
        rel.append(dm / de - 1); absd.append(dm - de); ang.append(a); ff.append(face_first)
    rel, absd, ang, ff = np.array(rel), np.array(absd), np.array(ang), np.array(ff)
    report(rel, absd, ang, ff, n_pen, missed)
    assert n_pen > 500 and missed == 0
    check_mpr_bounds(rel, absd, ang, ff, 2e-3)
