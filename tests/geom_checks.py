"""Independent geometry for the narrowphase tests: nothing here comes from oracle/hsr_oracle.c or the kernels.
  * brute-force SAT of two oriented boxes over the 15 candidate axes (overlap along every axis),
  * the EXACT penetration depth of two convex shapes: the distance from the origin to the nearest face of the convex hull of their
    Minkowski difference (scipy.spatial.ConvexHull), and that face's normal,
  * placements of the scene's geoms from link poses and the model tables, samplers of penetrating pairs.
Used on the oracle's contacts (tests/test_narrowphase_geometry.py, CPU) and on the HIP path's through the C-ABI (tests/test_gpu_geometry.py)."""
import numpy as np
from scipy.spatial import ConvexHull

GEOM_CYLINDER, GEOM_BOX, GEOM_MESH = 5, 6, 7


def quat2mat(q):
    w, x, y, z = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def random_quat(rng):
    q = rng.normal(size=4)
    return q / np.linalg.norm(q)


def geom_pose(m, xpos, xmat, g):
    """world position and orientation of geom g from its link's pose"""
    l = int(m.geom_link[g])
    R = xmat[l].reshape(3, 3)
    return xpos[l] + R @ m.geom_pos[g], R @ quat2mat(m.geom_quat[g])


def box_axes_overlaps(c1, R1, h1, c2, R2, h2):
    """the 15 SAT axes (unit, pointing from box 1 to box 2) of two oriented boxes and the overlap along each (negative: separated)"""
    d = c2 - c1
    axes = [R1[:, i] for i in range(3)] + [R2[:, j] for j in range(3)]
    kinds = [("face1", i) for i in range(3)] + [("face2", j) for j in range(3)]
    for i in range(3):
        for j in range(3):
            L = np.cross(R1[:, i], R2[:, j])
            n = np.linalg.norm(L)
            if n < 1e-6:
                continue
            axes.append(L / n); kinds.append(("edge", 3 * i + j))
    out = []
    for a, k in zip(axes, kinds):
        ra = float(np.abs(R1.T @ a) @ h1); rb = float(np.abs(R2.T @ a) @ h2)
        t = float(d @ a)
        out.append((a if t >= 0 else -a, k, ra + rb - abs(t)))
    return out


def inside_box(p, c, R, h, slack):
    return bool(np.all(np.abs(R.T @ (p - c)) <= h + slack))


def shape_vertices(m, g, pos, mat, nseg=720):
    """world vertices whose convex hull is geom g (the cylinder as a prism of nseg sides: its radius is exact to r (1 - cos(pi / nseg)) = 1e-5 r)"""
    t = int(m.geom_type[g])
    if t == GEOM_BOX:
        h = m.geom_size[g]
        loc = np.array([[sx * h[0], sy * h[1], sz * h[2]] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)])
    elif t == GEOM_MESH:
        a, n = int(m.geom_meshadr[g]), int(m.geom_meshnum[g])
        loc = m.mesh_vert[a:a + n]
    elif t == GEOM_CYLINDER:
        r, hh = m.geom_size[g][0], m.geom_size[g][1]
        ang = np.linspace(0, 2 * np.pi, nseg, endpoint=False)
        ring = np.stack([r * np.cos(ang), r * np.sin(ang)], 1)
        loc = np.concatenate([np.c_[ring, np.full(nseg, hh)], np.c_[ring, np.full(nseg, -hh)]])
    else:
        raise ValueError(t)
    return pos + loc @ mat.T


def exact_penetration(va, vb):
    """(depth, direction) of two intersecting convex vertex sets: the smallest translation of B that separates them is depth along direction
    (unit, from A towards B); depth < 0: they do not intersect (then -depth is a lower bound of the distance along the best face normal)."""
    md = (vb[None, :, :] - va[:, None, :]).reshape(-1, 3)          # B - A: contains the origin iff the shapes intersect
    hull = ConvexHull(md)
    n, off = hull.equations[:, :3], hull.equations[:, 3]             # n . x + off <= 0 inside
    dist = -off                                                      # distance of the origin to each face plane (positive: inside that face)
    k = int(np.argmin(dist))
    # B - A moves with B.  The origin is on the boundary once the hull is shifted by -dist_k n_k (the nearest boundary point is dist_k n_k): the
    # smallest separating translation of B is dist_k along -n_k - B is pushed away from A along the contact normal, which therefore is -n_k
    # (1-D check: A = [0, 1], B = [0.9, 2]: B - A = [-0.1, 2], nearest face at -0.1 with outward normal -1, B moves by +0.1).
    return float(dist[k]), -n[k]


def support(verts, d):
    return verts[int(np.argmax(verts @ d))]
