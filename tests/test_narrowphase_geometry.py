"""The narrowphase's GEOMETRY checked against constructions that share no code with it (CPU, the oracle's
contacts; tests/test_gpu_geometry.py asks the same of the HIP path through the C-ABI).  Box-box and MPR are the two routines behind
`self.sim.step()` (hsr/env.py:123 -> mj_collision) that oracle and kernels restate from their purpose: parity between the two
cannot see an error they share.
  * box <-> box (the block on the pan, blocks against each other: hsr/util.py:115-125, hsr/models/world.xml:83-84): brute-force SAT over the 15
    axes; the reported normal is one of them and its overlap is the smallest one up to the routine's documented preference for face axes
    (an edge axis has to be 5 % better); no point is deeper than that overlap, the deepest incident vertex is reported with exactly that depth
    when it lies over the reference face; every point lies in both boxes grown by half its depth; the normal points from geom 1 to geom 2;
    separated boxes (some axis without overlap) yield nothing.
  * convex <-> box (the robot's 14 hulls and the wrist cylinder against a block: hsr/models/hsr.mjcf:180,192,217,228 ...): MPR's depth and
    direction against the EXACT penetration depth - the distance of the origin to the nearest face of the convex hull of the Minkowski
    difference - at penetrations of 0.1 ... 5 mm.  libccd's MPR (which MuJoCo shares) measures along the portal it ends on, not the minimum
    over all directions: the test reports the distribution and bounds it."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from hsr_env_amd.compiler import load_config          # noqa: E402
from oracle.oracle import OracleSim                    # noqa: E402
import geom_checks as gc                               # noqa: E402


def boxbox_states(m, n, rng):
    """qpos rows of cfg4 in which block 0 and block 1 overlap at random relative poses (a third of them resting flat on each other or edge
    on face), block 2 lies on / in the pan at a random pose; the robot stays at qpos0."""
    q = np.tile(m.qpos0, (n, 1))
    fa = [int(a) for a in m.free_joint_qadrs()]
    h = m.geom_size[17]
    for e in range(n):
        kind = e % 3
        qa = gc.random_quat(rng) if kind != 1 else np.array([1.0, 0, 0, 0])
        pa = np.array([0.0, 0.0, 0.8]) + rng.uniform(-0.02, 0.02, 3)
        Ra = gc.quat2mat(qa)
        if kind == 0:          # anything: random relative rotation, centres closer than the boxes' extent
            qb = gc.random_quat(rng)
            pb = pa + rng.uniform(-1, 1, 3) * np.array([0.07, 0.04, 0.03])
        elif kind == 1:        # resting: same orientation up to a yaw, shifted within the face, sunk by up to 2 mm
            yaw = rng.uniform(-np.pi, np.pi)
            qb = np.array([np.cos(yaw / 2), 0, 0, np.sin(yaw / 2)])
            pb = pa + np.array([rng.uniform(-0.04, 0.04), rng.uniform(-0.02, 0.02), 2 * h[2] - rng.uniform(1e-5, 2e-3)])
        else:                  # edge / corner first: a tilted block pushed into a face
            qb = gc.random_quat(rng)
            Rb = gc.quat2mat(qb)
            reach = float(np.abs(Rb.T @ Ra[:, 2]) @ h)          # extent of block b along a's face normal
            pb = pa + Ra @ np.array([rng.uniform(-0.03, 0.03), rng.uniform(-0.015, 0.015), h[2] + reach - rng.uniform(1e-5, 3e-3)])
        q[e, fa[0]:fa[0] + 3] = pa; q[e, fa[0] + 3:fa[0] + 7] = qa
        q[e, fa[1]:fa[1] + 3] = pb; q[e, fa[1] + 3:fa[1] + 7] = qb
        qc = gc.random_quat(rng) if e % 2 else np.array([1.0, 0, 0, 0])
        Rc = gc.quat2mat(qc)
        reach = float(np.abs(Rc.T @ np.array([0, 0, 1.0])) @ h)
        q[e, fa[2]:fa[2] + 3] = [rng.uniform(-0.1, 0.1), rng.uniform(-0.2, 0.2), 0.405 + reach - rng.uniform(1e-5, 3e-3)]
        q[e, fa[2] + 3:fa[2] + 7] = qc
    return q


def check_boxbox(m, xpos, xmat, contacts, g1, g2, tol, stats):
    """contacts: rows (pos3, normal3, dist) of the pair (g1, g2); tol: absolute tolerance of the arithmetic that produced them"""
    c1, R1 = gc.geom_pose(m, xpos, xmat, g1); c2, R2 = gc.geom_pose(m, xpos, xmat, g2)
    h1, h2 = m.geom_size[g1], m.geom_size[g2]
    sat = gc.box_axes_overlaps(c1, R1, h1, c2, R2, h2)
    omin = min(o for _, _, o in sat)
    if omin < -tol:
        assert len(contacts) == 0, ("separated boxes with contacts", omin)
        stats["separated"] += 1
        return
    if len(contacts) == 0:
        stats["touching_without_points" if omin < 10 * tol else "overlap_without_points"] += 1
        return
    n = contacts[0, 3:6]
    assert np.abs(contacts[:, 3:6] - n).max() <= 10 * tol                      # one normal for the whole manifold
    match = [(a, k, o) for a, k, o in sat if a @ n > 1 - max(1e-9, 10 * tol)]
    assert match, ("normal is none of the 15 axes", n)
    a, kind, on = max(match, key=lambda t: float(t[0] @ n))          # (a face axis and an edge axis can lie within the fp32 tolerance of each other: the closest one)
    assert n @ (c2 - c1) >= -tol
    # the routine prefers face axes: an edge axis must beat the best face axis by 5 %; among the face axes the first strictly better wins
    oface = min(o for _, k, o in sat if k[0] != "edge")
    if kind[0] == "edge":
        assert on * 1.05 <= oface + 1e-9 + 10 * tol, (on, oface)
        # (the nine edge axes are tried in order and each has to beat the running best by the same 5 %: the one kept is within 5 % of the smallest)
        assert on <= 1.05 * min(o for _, k, o in sat if k[0] == "edge") + 1e-9 + 10 * tol
        stats["edge"] += 1
    else:
        assert on <= oface + 10 * tol, (on, oface)
        assert omin >= on / 1.05 - 1e-9 - 10 * tol, ("an edge axis was better by more than the preference", on, omin)
        stats["face"] += 1
    depth = -contacts[:, 6]
    assert depth.min() > -tol and depth.max() <= on + 10 * tol, (depth, on)
    for p, d in zip(contacts[:, 0:3], depth):
        ok = gc.inside_box(p, c1, R1, h1, 0.5 * d + 10 * tol) and gc.inside_box(p, c2, R2, h2, 0.5 * d + 10 * tol)
        if kind[0] == "edge":
            # an edge contact is the midpoint of the closest points of the two edge LINES; where those fall beyond the ends of the edges (the
            # boxes then touch corner to edge) the point lies outside: counted, not asserted
            stats["edge_point_outside"] += 0 if ok else 1
        else:
            assert ok, (kind, p, d)
    if kind[0] == "edge":
        assert len(contacts) == 1 and abs(depth[0] - on) <= 10 * tol
    else:
        # the deepest vertex of the incident box along the reference normal: reported with the full overlap when it lies over the reference face
        ref1 = kind[0] == "face1"
        cr, Rr, hr, ci, Ri, hi = (c1, R1, h1, c2, R2, h2) if ref1 else (c2, R2, h2, c1, R1, h1)
        nref = a if ref1 else -a
        v = ci + Ri @ (-np.sign(Ri.T @ nref) * hi)
        loc = Rr.T @ (v - cr)
        ax = kind[1]
        side = [i for i in range(3) if i != ax]
        if all(abs(loc[i]) <= hr[i] - 1e-6 for i in side) and np.abs(Ri.T @ nref).min() > 1e-3:
            assert abs(depth.max() - on) <= 10 * tol, (depth, on)
            stats["deepest_vertex_reported"] += 1


def test_box_box_against_brute_force_sat():
    m = load_config("cfg4")
    rng = np.random.default_rng(5)
    n = 3400
    q = boxbox_states(m, n, rng)
    o = OracleSim(m)
    stats = dict(separated=0, face=0, edge=0, touching_without_points=0, overlap_without_points=0, deepest_vertex_reported=0, edge_point_outside=0)
    for e in range(n):
        o.qpos[:] = q[e]; o.qvel[:] = 0
        o.forward()
        oc = o.contacts()
        for g1, g2 in ((17, 18), (1, 19)):
            rows = oc[(oc[:, 13] == g1) & (oc[:, 14] == g2)] if len(oc) else np.zeros((0, 17))
            check_boxbox(m, o.xpos, o.xmat.reshape(-1, 3, 3), np.c_[rows[:, 0:3], rows[:, 3:6], rows[:, 12]], g1, g2, 1e-10, stats)
    print(stats)
    assert stats["face"] > 3000 and stats["edge"] > 300 and stats["deepest_vertex_reported"] > 1500
    # boxes that overlap (every axis) and get no point: the clipped incident face has no vertex below the reference face - rare, counted
    assert stats["overlap_without_points"] <= 0.01 * 2 * n, stats


def mat2quat(R):
    w = np.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
    if w > 1e-6:
        return np.array([w, (R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w)])
    x = np.sqrt(max(0.0, 1 + R[0, 0] - R[1, 1] - R[2, 2])) / 2
    if x > 1e-6:
        return np.array([(R[2, 1] - R[1, 2]) / (4 * x), x, (R[0, 1] + R[1, 0]) / (4 * x), (R[0, 2] + R[2, 0]) / (4 * x)])
    y = np.sqrt(max(0.0, 1 - R[0, 0] + R[1, 1] - R[2, 2])) / 2
    if y > 1e-6:
        return np.array([(R[0, 2] - R[2, 0]) / (4 * y), (R[0, 1] + R[1, 0]) / (4 * y), y, (R[1, 2] + R[2, 1]) / (4 * y)])
    return np.array([0.0, 0.0, 0.0, 1.0])


def mpr_samples(m, rng, per_pair):
    """(pair geoms (box first), qpos) with the block pushed 0.1 ... 5 mm into one of the robot's convex geoms along a random direction"""
    o = OracleSim(m)
    o.qpos[:] = m.qpos0; o.forward()
    xpos, xmat = o.xpos.copy(), o.xmat.reshape(-1, 3, 3).copy()
    fa = int(m.free_joint_qadrs()[0])
    hb = m.geom_size[17]
    out = []
    for g in range(2, 17):
        pos, mat = gc.geom_pose(m, xpos, xmat, g)
        verts = gc.shape_vertices(m, g, pos, mat)
        for _ in range(per_pair):
            u = rng.normal(size=3); u /= np.linalg.norm(u)
            s = gc.support(verts, u)
            delta = 10 ** rng.uniform(-4, np.log10(5e-3))
            face_first = rng.random() >= 0.5
            if not face_first:              # a corner of the block first: its deepest vertex lies delta below the hull's support point along u
                qb = gc.random_quat(rng)
                Rb = gc.quat2mat(qb)
                cb = s - delta * u + Rb @ (np.sign(Rb.T @ u) * hb)
            else:                           # a face of the block first: the face with normal -u, the support point somewhere over its middle half
                z = u
                x = np.cross(z, rng.normal(size=3)); x /= np.linalg.norm(x)
                Rb = np.stack([x, np.cross(z, x), z], 1)
                qb = mat2quat(Rb)
                cb = s + u * (hb[2] - delta) + Rb @ np.array([rng.uniform(-0.5, 0.5) * hb[0], rng.uniform(-0.5, 0.5) * hb[1], 0.0])
            q = m.qpos0.copy()
            q[fa:fa + 3] = cb; q[fa + 3:fa + 7] = qb
            out.append((g, q, face_first))
    return out, xpos, xmat


def mpr_compare(m, xpos, xmat, g, q, contact):
    """contact: (pos3, normal3 from geom1 to geom2, dist) of the pair {block 17, geom g} or None; returns (exact depth, mpr depth, angle)"""
    fa = int(m.free_joint_qadrs()[0])
    pb, Rb = q[fa:fa + 3], gc.quat2mat(q[fa + 3:fa + 7])
    vb = gc.shape_vertices(m, 17, pb, Rb)
    pos, mat = gc.geom_pose(m, xpos, xmat, g)
    vg = gc.shape_vertices(m, g, pos, mat)
    first_is_block = (17, g) in set(zip([int(a) for a in m.pair_geom1], [int(b) for b in m.pair_geom2]))
    va, vb2 = (vb, vg) if first_is_block else (vg, vb)
    depth, nrm = gc.exact_penetration(va, vb2)
    if contact is None:
        return depth, None, None
    return depth, -contact[6], float(np.degrees(np.arccos(np.clip(contact[3:6] @ nrm, -1, 1))))


def test_mpr_depth_against_the_exact_penetration_depth():
    m = load_config("cfg3")
    rng = np.random.default_rng(9)
    samples, xpos, xmat = mpr_samples(m, rng, 40)
    o = OracleSim(m)
    rel, absd, ang, ff, missed, n_pen = [], [], [], [], 0, 0
    for g, q, face_first in samples:
        o.qpos[:] = q; o.qvel[:] = 0
        o.forward()
        oc = o.contacts()
        rows = oc[((oc[:, 13] == 17) & (oc[:, 14] == g)) | ((oc[:, 13] == g) & (oc[:, 14] == 17))] if len(oc) else np.zeros((0, 17))
        con = np.r_[rows[0, 0:3], rows[0, 3:6], rows[0, 12]] if len(rows) else None
        de, dm, a = mpr_compare(m, xpos, xmat, g, q, con)
        if de <= 1e-7:
            assert con is None or -con[6] < 1e-5
            continue
        n_pen += 1
        if con is None:
            missed += 1
            continue
        assert dm >= de - 2e-6, ("MPR below the exact depth", g, de, dm)          # no direction separates with less than the exact depth (MPR stops 1e-6 short of the surface: its tolerance)
        rel.append(dm / de - 1); absd.append(dm - de); ang.append(a); ff.append(face_first)
    rel, absd, ang, ff = np.array(rel), np.array(absd), np.array(ang), np.array(ff)
    report(rel, absd, ang, ff, n_pen, missed)
    assert n_pen > 500 and missed == 0
    check_mpr_bounds(rel, absd, ang, ff, 0.0)


def report(rel, absd, ang, ff, n_pen, missed):
    for nm, sel in (("a face of the block first", ff), ("a corner of the block first", ~ff)):
        r, a, g = rel[sel], absd[sel], ang[sel]
        print(f"{nm}: {sel.sum()} of {n_pen} penetrating samples ({missed} without a contact); MPR depth / exact - 1: median {np.median(r):.1e} p90 {np.percentile(r, 90):.1e} "
              f"p99 {np.percentile(r, 99):.1e} max {r.max():.1e}; excess p90 {np.percentile(a, 90):.1e} max {a.max():.1e} m; direction median {np.median(g):.2f} p90 {np.percentile(g, 90):.1f} max {g.max():.1f} deg")


def check_mpr_bounds(rel, absd, ang, ff, slack):
    """What was measured (fp64 oracle, 600 samples at exact depths of 0.1 ... 5 mm; `-s` prints it), with a margin; slack: what fp32 adds.
    A face of the block against the hull: the portal ends on that face and the depth is exact (relative excess: median 6e-16, p90 8e-12,
    p99 4.5e-2, max 0.43 - the last per cent are faces that meet a hull edge obliquely).  A corner of the block first: libccd's MPR refines the
    portal that the ray from the geoms' centres hits and measures the depth THERE - an upper bound of the exact depth (asserted per sample),
    right in the median (2.6e-3 above, direction 6.8 deg off) and p90 19 %, p99 48 %, at worst 97 % too deep (directions up to 79 deg off the
    exact one) where the corner enters obliquely; absolute excess p90 1.4e-4 m, max 1.4e-3 m.  MuJoCo shares the routine (mjc_Convex ->
    ccdMPRPenetration): this is the approximation the reference itself runs on, not a defect of the restatement."""
    f, c = rel[ff], rel[~ff]
    assert np.median(f) < 1e-6 + slack and np.percentile(f, 90) < 1e-3 + slack and f.max() < 0.8, (np.median(f), np.percentile(f, 90), f.max())
    assert np.median(c) < 2e-2 + slack and np.percentile(c, 90) < 0.35 and c.max() < 1.5, (np.median(c), np.percentile(c, 90), c.max())
    assert np.percentile(absd, 90) < 3e-4 + slack and absd.max() < 3e-3
    assert np.median(ang[ff]) < 0.1 + 1e3 * slack and np.median(ang[~ff]) < 15.0
