"""Offline model compiler: HSR MJCF (+ STL meshes) -> flat constant tables ("model blob").

This is the build's counterpart of ``mujoco_py.load_model_from_path`` applied to the XML that
``hsr.util.mutate_xml`` would have produced (reference: hsr/mujoco_env.py:33-34,
hsr/util.py:87-182).  It runs once, on the host, in fp64, and needs the reference's *data*
files (hsr/models/world.xml, hsr/models/hsr.mjcf, hsr/hsr_meshes/meshes/**/*.stl); the GPU box
only ever sees the compiled blobs committed under ``hsr_env_amd/models/``.

What it does (decisions H1-H7 of SURVEY.md §7 are recorded in ``Model.meta``):
  * applies the reference's XML mutations directly: block injection (util.py:106-127) and the
    ``--use-dof`` actuator/joint filter (util.py:137-146);
  * resolves defaults (second top-level ``<default class="all">`` applies to all geoms, H3),
    ``angle="degree"`` (H4), ``inertiafromgeom="true"`` (H2), malformed ``pos`` (H1),
    unnormalised quaternions (H5);
  * folds every joint-less body into its nearest jointed ancestor ("link" = MuJoCo weld body),
    so the device tables hold <= 1 + 5 + n_blocks rigid links instead of ~50 bodies;
  * computes convex hulls of collidable meshes, geom bounding spheres, the static candidate
    geom-pair list after MuJoCo's filters (same weld body, parent-child weld bodies unless the
    parent is the world, <exclude>, contype/conaffinity);
  * computes qpos0 statistics used by the constraint regulariser (body/dof invweight0,
    meaninertia).

Engine semantics are restated from MuJoCo's public documentation (the engine itself is absent
from the reference tree and from this container) - parity with mujoco-py is
therefore unpinned; see DESIGN.md.
"""
from __future__ import annotations

import json
import struct
import xml.etree.ElementTree as ET
from dataclasses import dataclass, field
from pathlib import Path
from typing import Dict, List, Optional, Sequence

import numpy as np

# MuJoCo geom type enum values (kept so tables read like mjModel)
GEOM_PLANE, GEOM_SPHERE, GEOM_CYLINDER, GEOM_BOX, GEOM_MESH = 0, 2, 5, 6, 7
GEOM_TYPES = {"plane": GEOM_PLANE, "sphere": GEOM_SPHERE, "cylinder": GEOM_CYLINDER,
              "box": GEOM_BOX, "mesh": GEOM_MESH}
# dof types
DOF_SLIDE, DOF_HINGE, DOF_FREE_LIN, DOF_FREE_ANG = 0, 1, 2, 3
# narrowphase function per candidate pair
FN_PLANE_BOX, FN_PLANE_CONVEX, FN_BOX_BOX, FN_CONVEX = 0, 1, 2, 3
UNLIMITED = 1e30     # range of an actuator without ctrllimited / forcelimited (finite in fp32)
FN_MAXCON = {FN_PLANE_BOX: 4, FN_PLANE_CONVEX: 1, FN_BOX_BOX: 8, FN_CONVEX: 1}

DEFAULT_REF_ROOT = Path("/root/reference/hsr")
ALL_DOFS = ["slide_x", "slide_y", "arm_lift_joint", "arm_flex_joint", "wrist_roll_joint",
            "hand_l_proximal_joint", "hand_r_proximal_joint"]

BLOB_MAGIC = b"HSRM0001"


# ----------------------------------------------------------------------------- math helpers
def quat_normalize(q):
    q = np.asarray(q, dtype=np.float64)
    return q / np.linalg.norm(q)


def quat_mul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([aw * bw - ax * bx - ay * by - az * bz,
                     aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw])


def quat_to_mat(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def axis_angle_quat(axis, angle):
    axis = np.asarray(axis, dtype=np.float64)
    s = np.sin(0.5 * angle)
    return np.array([np.cos(0.5 * angle), axis[0] * s, axis[1] * s, axis[2] * s])


def _vec(text, n, default):
    """Parse an MJCF float vector; malformed text (H1: pos="0 0hsr") falls back to default."""
    if text is None:
        return np.array(default, dtype=np.float64)
    try:
        v = np.array([float(t) for t in text.split()], dtype=np.float64)
    except ValueError:
        return np.array(default, dtype=np.float64)
    if v.size != n:
        if v.size < n and n == 3 and v.size >= 1:   # geom size with fewer entries
            out = np.zeros(n)
            out[:v.size] = v
            return out
        return np.array(default, dtype=np.float64)
    return v


# ----------------------------------------------------------------------------- mesh handling
def load_stl(path: Path) -> np.ndarray:
    """Binary STL -> triangles [n,3,3] (fp64)."""
    raw = path.read_bytes()
    n = struct.unpack("<I", raw[80:84])[0]
    rec = np.dtype([("n", "<3f4"), ("v", "<9f4"), ("a", "<u2")])
    arr = np.frombuffer(raw, dtype=rec, count=n, offset=84)
    return arr["v"].reshape(n, 3, 3).astype(np.float64)


def mesh_inertia_legacy(tris: np.ndarray):
    """MuJoCo's legacy mesh inertia: pyramids from the surface centroid to every face, |volume|.

    Returns (volume, com[3], inertia about com [3,3]) for unit density.
    """
    a, b, c = tris[:, 0], tris[:, 1], tris[:, 2]
    area = 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1)
    center = ((a + b + c) / 3.0 * area[:, None]).sum(0) / area.sum()
    a0, b0, c0 = a - center, b - center, c - center
    vol = np.abs(np.einsum("ij,ij->i", a0, np.cross(b0, c0))) / 6.0
    V = vol.sum()
    # tetra (0,a0,b0,c0): centroid (a0+b0+c0)/4 ; second moment about the apex:
    # int x x^T dV = vol/20 * (sum_i v_i v_i^T + (sum v)(sum v)^T)
    cen = (a0 + b0 + c0) / 4.0
    com0 = (cen * vol[:, None]).sum(0) / V
    s = a0 + b0 + c0
    P = (np.einsum("n,ni,nj->ij", vol, a0, a0) + np.einsum("n,ni,nj->ij", vol, b0, b0) +
         np.einsum("n,ni,nj->ij", vol, c0, c0) + np.einsum("n,ni,nj->ij", vol, s, s)) / 20.0
    # covariance about the pyramid apex (center) -> about com
    P -= V * np.outer(com0, com0)
    inertia = np.trace(P) * np.eye(3) - P
    return V, center + com0, inertia


def convex_hull_vertices(points: np.ndarray) -> np.ndarray:
    from scipy.spatial import ConvexHull
    uniq = np.unique(points.round(9), axis=0)
    hull = ConvexHull(uniq)
    return uniq[np.sort(hull.vertices)]


# ----------------------------------------------------------------------------- MJCF records
@dataclass
class _Geom:
    name: str
    type: int
    size: np.ndarray
    pos: np.ndarray
    quat: np.ndarray
    mesh: Optional[str]
    contype: int
    conaffinity: int
    condim: int
    friction: np.ndarray
    solref: np.ndarray
    solimp: np.ndarray
    mass: Optional[float]
    density: float


@dataclass
class _Joint:
    name: str
    type: str            # slide | hinge | free
    axis: np.ndarray
    pos: np.ndarray
    limited: bool
    range: np.ndarray
    damping: float


@dataclass
class _Body:
    name: str
    pos: np.ndarray
    quat: np.ndarray
    mocap: bool
    parent: int
    joints: List[_Joint] = field(default_factory=list)
    geoms: List[_Geom] = field(default_factory=list)
    inertial: Optional[dict] = None


def _apply_setters(root, set_xml):
    """hsr/util.py:129-135 (`for change in changes`): the path's last component is the attribute, the rest an ElementTree
    path relative to the root of the file being mutated; a path that matches nothing in this file is skipped."""
    import re
    for path, value in set_xml:
        parent = re.sub('/[^/]*$', '', str(path))
        elt = root.find(parent)
        if isinstance(elt, ET.Element):
            elt.set(re.search('[^/]*$', str(path))[0], str(value))


def _parse_tree(ref_root: Path, xml_file: str, dofs: Sequence[str], n_blocks: int,
                block_pos: np.ndarray, set_xml=(), block_geom=None):
    xml_path = ref_root / xml_file
    root = ET.parse(xml_path).getroot()
    meta = {}
    # -- block injection (util.py:106-127), then the --set-xml changes (util.py:129-135), file by file as mutate_tree does
    worldbody = root.find("worldbody")
    for i in range(n_blocks):
        name = f"block{i}"
        body = ET.SubElement(worldbody, "body",
                             attrib=dict(name=name, pos=" ".join(repr(float(x)) for x in block_pos[i])))
        gattr = dict(name=name, type="box", mass="1", size=".05 .025 .017", condim="6", solimp="0.99 0.99 0.01", solref="0.01 1")
        if block_geom:              # test scenes only: another shape for the injected body (e.g. one of the robot's hulls as a free body)
            gattr.update(block_geom)
            if gattr.get("type") == "mesh":
                gattr.pop("size", None)
        ET.SubElement(body, "geom", attrib=gattr)
        ET.SubElement(body, "freejoint", attrib=dict(name=f"block{i}joint"))
    _apply_setters(root, set_xml)

    # -- options / compiler -------------------------------------------------------------
    opt = {"timestep": 0.002, "impratio": 1.0, "cone": "pyramidal"}
    for o in root.findall("option"):
        for k in ("timestep", "impratio"):
            if o.get(k) is not None:
                opt[k] = float(o.get(k))
        if o.get("cone") is not None:
            opt["cone"] = o.get("cone")
    comp = root.find("compiler")
    degree = comp.get("angle", "degree") == "degree"
    inertiafromgeom = comp.get("inertiafromgeom", "auto") == "true"
    meshdir = (xml_path.parent / comp.get("meshdir", ".")).resolve()
    size = root.find("size")
    opt["njmax"] = int(size.get("njmax", 500))
    opt["nconmax"] = int(size.get("nconmax", 100))

    # -- defaults (H3) --------------------------------------------------------------------
    geom_global: Dict[str, str] = {}
    geom_class: Dict[str, Dict[str, str]] = {}
    for top in root.findall("default"):
        for g in top.findall("geom"):
            geom_global.update(g.attrib)
        for sub in top.findall("default"):
            for g in sub.findall("geom"):
                geom_class.setdefault(sub.get("class"), {}).update(g.attrib)

    meshes = {m.get("name"): meshdir / m.get("file") for m in root.find("asset").findall("mesh")}

    # -- splice <include> (util.py:148-151 keeps includes relative) -----------------------
    for i, child in enumerate(list(worldbody)):
        if child.tag == "include":
            inc = ET.parse(xml_path.parent / child.get("file")).getroot()
            _apply_setters(inc, set_xml)
            worldbody.remove(child)
            for j, b in enumerate(list(inc)):
                worldbody.insert(i + j, b)

    # -- DOF filter (util.py:137-146) ---------------------------------------------------------
    actuators = []
    for acts in root.iter("actuator"):
        for a in list(acts):
            if a.get("joint") in dofs:
                actuators.append(a)
    for body in root.iter("body"):
        for j in body.findall("joint"):
            if j.get("name") not in dofs:
                body.remove(j)

    excludes = [(e.get("body1"), e.get("body2")) for c in root.findall("contact")
                for e in c.findall("exclude")]

    # -- walk bodies -------------------------------------------------------------------------
    bodies: List[_Body] = [_Body("world", np.zeros(3), np.array([1., 0, 0, 0]), False, -1)]

    def parse_geom(g, idx):
        at = dict(geom_global)
        at.update(geom_class.get(g.get("class"), {}))
        at.update(g.attrib)
        gtype = GEOM_TYPES[at.get("type", "sphere")]
        solimp = np.array([0.9, 0.95, 0.001, 0.5, 2.0])
        if "solimp" in at:
            v = [float(t) for t in at["solimp"].split()]
            solimp[:len(v)] = v
        fr = np.array([1.0, 0.005, 0.0001])
        if "friction" in at:
            v = [float(t) for t in at["friction"].split()]
            fr[:len(v)] = v
        return _Geom(name=at.get("name", f"geom{idx}"), type=gtype,
                     size=_vec(at.get("size"), 3, [0, 0, 0]),
                     pos=_vec(at.get("pos"), 3, [0, 0, 0]),
                     quat=quat_normalize(_vec(at.get("quat"), 4, [1, 0, 0, 0])),
                     mesh=at.get("mesh"), contype=int(at.get("contype", 1)),
                     conaffinity=int(at.get("conaffinity", 1)), condim=int(at.get("condim", 3)),
                     friction=fr, solref=_vec(at.get("solref"), 2, [0.02, 1.0]), solimp=solimp,
                     mass=float(at["mass"]) if "mass" in at else None,
                     density=float(at.get("density", 1000.0)))

    ngeom_seen = [0]

    def walk(elem, parent_id):
        for g in elem.findall("geom"):
            bodies[parent_id].geoms.append(parse_geom(g, ngeom_seen[0]))
            ngeom_seen[0] += 1
        for b in elem.findall("body"):
            body = _Body(name=b.get("name"), pos=_vec(b.get("pos"), 3, [0, 0, 0]),
                         quat=quat_normalize(_vec(b.get("quat"), 4, [1, 0, 0, 0])),
                         mocap=b.get("mocap", "false") == "true", parent=parent_id)
            if b.get("pos") is not None and _vec(b.get("pos"), 3, [np.nan] * 3)[0] != body.pos[0]:
                meta.setdefault("H1_malformed_pos", []).append(b.get("name"))
            for j in b.findall("joint"):
                jt = j.get("type", "hinge")
                rng = _vec(j.get("range"), 2, [0, 0])
                if jt == "hinge" and degree:
                    rng = np.deg2rad(rng)
                bodies_j = _Joint(name=j.get("name"), type=jt,
                                  axis=quat_normalize(_vec(j.get("axis"), 3, [0, 0, 1])),
                                  pos=_vec(j.get("pos"), 3, [0, 0, 0]),
                                  limited=j.get("limited", "false") == "true", range=rng,
                                  damping=float(j.get("damping", 0.0)))
                body.joints.append(bodies_j)
            for j in b.findall("freejoint"):
                body.joints.append(_Joint(name=j.get("name"), type="free", axis=np.zeros(3),
                                          pos=np.zeros(3), limited=False, range=np.zeros(2),
                                          damping=0.0))
            inert = b.find("inertial")
            if inert is not None:
                body.inertial = dict(pos=_vec(inert.get("pos"), 3, [0, 0, 0]),
                                     quat=quat_normalize(_vec(inert.get("quat"), 4, [1, 0, 0, 0])),
                                     mass=float(inert.get("mass")),
                                     diag=_vec(inert.get("diaginertia"), 3, [0, 0, 0]))
            bodies.append(body)
            walk(b, len(bodies) - 1)

    walk(worldbody, 0)
    return dict(opt=opt, bodies=bodies, meshes=meshes, actuators=actuators, excludes=excludes,
                inertiafromgeom=inertiafromgeom, meta=meta)


# ----------------------------------------------------------------------------- the model
_ARRAY_FIELDS = [
    # scalars packed as arrays for a uniform container
    "sizes", "opt",
    "qpos0",
    "link_parent", "link_pos", "link_quat", "link_dofadr", "link_dofnum", "link_qposadr",
    "link_free", "link_mass", "link_com", "link_inertia", "link_dofmask",
    "dof_link", "dof_type", "dof_axis", "dof_pos", "dof_parent", "dof_damping", "dof_qposadr",
    "dof_invweight0", "dof_limited", "dof_range", "dof_solref", "dof_solimp",
    "body_link", "body_pos", "body_quat", "body_mocap",
    "geom_type", "geom_link", "geom_body", "geom_pos", "geom_quat", "geom_size", "geom_rbound",
    "geom_condim", "geom_meshadr", "geom_meshnum", "geom_invweight", "geom_aabb",
    "mesh_vert",
    "pair_geom1", "pair_geom2", "pair_fn", "pair_condim", "pair_slot", "pair_friction",
    "pair_solref", "pair_solimp",
    "act_dof", "act_gear", "act_kp", "act_ctrlrange", "act_forcerange",
]

# index constants into ``sizes`` / ``opt`` (mirrored in include/hsrsim.h and oracle/hsr_oracle.c)
SZ_NQ, SZ_NV, SZ_NU, SZ_NLINK, SZ_NBODY, SZ_NGEOM, SZ_NPAIR, SZ_NMESHVERT, SZ_NSLOT, \
    SZ_NLIMIT, SZ_NCONMAX, SZ_NJMAX, SZ_NMOCAP, SZ_NDENSE = range(14)
OPT_TIMESTEP, OPT_IMPRATIO, OPT_GRAV_Z, OPT_TOLERANCE, OPT_ITERATIONS, OPT_LS_ITERATIONS, \
    OPT_LS_TOLERANCE, OPT_MPR_TOLERANCE, OPT_MPR_ITERATIONS, OPT_MEANINERTIA = range(10)


@dataclass
class Model:
    arrays: Dict[str, np.ndarray]
    names: Dict[str, List[str]]
    meta: Dict[str, object]

    def __getattr__(self, k):
        arrays = object.__getattribute__(self, "arrays")
        if k in arrays:
            return arrays[k]
        raise AttributeError(k)

    # -- sizes
    @property
    def nq(self): return int(self.arrays["sizes"][SZ_NQ])
    @property
    def nv(self): return int(self.arrays["sizes"][SZ_NV])
    @property
    def nu(self): return int(self.arrays["sizes"][SZ_NU])
    @property
    def nlink(self): return int(self.arrays["sizes"][SZ_NLINK])
    @property
    def nbody(self): return int(self.arrays["sizes"][SZ_NBODY])
    @property
    def ngeom(self): return int(self.arrays["sizes"][SZ_NGEOM])
    @property
    def npair(self): return int(self.arrays["sizes"][SZ_NPAIR])
    @property
    def nslot(self): return int(self.arrays["sizes"][SZ_NSLOT])
    @property
    def timestep(self): return float(self.arrays["opt"][OPT_TIMESTEP])

    def body_id(self, name: str) -> int:
        return self.names["body"].index(name)

    def scalar_joints(self):
        """(qpos addresses, dof addresses) of the 1-dof joints (the robot), in joint order; scenes differ in whether the
        block's free joint comes before (cupboard-world.xml) or after (world.xml + util.py injection) the robot."""
        qa = [a for (a, n) in self.meta["joint_qposadr"] if n == 1]
        da = [d for (a, n), d in zip(self.meta["joint_qposadr"], self.meta["joint_dofadr"]) if n == 1]
        return np.array(qa, dtype=int), np.array(da, dtype=int)

    def free_joint_qadrs(self):
        """qpos start address of every free joint (x y z qw qx qy qz), in joint order."""
        return [a for (a, n) in self.meta["joint_qposadr"] if n == 7]

    def block_body(self) -> str:
        """Name of the first free body: `block0` (util.py:109) or `block` (cupboard-world.xml:113)."""
        for cand in ("block0", "block"):
            if cand in self.names["body"]:
                return cand
        return ""

    def joint_qpos_addr(self, name: str):
        """mujoco_py ``model.get_joint_qpos_addr`` (reference use: hsr/env.py:153)."""
        j = self.names["joint"].index(name)
        adr, n = self.meta["joint_qposadr"][j]
        return adr if n == 1 else (adr, adr + n)

    # -- (de)serialisation ------------------------------------------------------------------
    def to_bytes(self) -> bytes:
        """Container: magic | u32 n | n x (name[32], u32 dtype, u32 ndim, u32 shape[4], u64 off,
        u64 nbytes) | json_len u64 | json | data (8-byte aligned).  dtype 0=f64, 1=i32."""
        entries, blobs, off = [], [], 0
        for name in _ARRAY_FIELDS:
            a = self.arrays[name]
            if a.dtype.kind == "f":
                a = np.ascontiguousarray(a, dtype="<f8"); code = 0
            else:
                a = np.ascontiguousarray(a, dtype="<i4"); code = 1
            shape = list(a.shape) + [0] * (4 - a.ndim)
            raw = a.tobytes()
            pad = (-len(raw)) % 8
            entries.append(struct.pack("<32sII4IQQ", name.encode(), code, a.ndim, *shape, off, len(raw)))
            blobs.append(raw + b"\0" * pad)
            off += len(raw) + pad
        js = json.dumps(dict(names=self.names, meta=self.meta)).encode()
        js += b" " * ((-len(js)) % 8)
        head = BLOB_MAGIC + struct.pack("<I", len(entries)) + b"\0" * 4
        return head + b"".join(entries) + struct.pack("<Q", len(js)) + js + b"".join(blobs)

    @staticmethod
    def from_bytes(raw: bytes) -> "Model":
        assert raw[:8] == BLOB_MAGIC, "not an HSRM blob"
        n = struct.unpack("<I", raw[8:12])[0]
        p = 16
        ents = []
        esz = struct.calcsize("<32sII4IQQ")
        for _ in range(n):
            ents.append(struct.unpack("<32sII4IQQ", raw[p:p + esz])); p += esz
        jl = struct.unpack("<Q", raw[p:p + 8])[0]; p += 8
        js = json.loads(raw[p:p + jl].decode()); p += jl
        arrays = {}
        for name, code, ndim, s0, s1, s2, s3, off, nb in ents:
            shape = (s0, s1, s2, s3)[:ndim]
            dt = "<f8" if code == 0 else "<i4"
            arrays[name.rstrip(b"\0").decode()] = np.frombuffer(
                raw, dtype=dt, count=nb // (8 if code == 0 else 4), offset=p + off).reshape(shape).copy()
        return Model(arrays=arrays, names=js["names"], meta=js["meta"])

    def save(self, path):
        Path(path).write_bytes(self.to_bytes())

    @staticmethod
    def load(path) -> "Model":
        return Model.from_bytes(Path(path).read_bytes())


# ----------------------------------------------------------------------------- numpy reference
def link_kinematics(m: Model, qpos: np.ndarray):
    """fp64 forward kinematics over links -> (xpos[nlink,3], xquat[nlink,4]).

    Restates mj_kinematics for the folded tree: body frame = parent * (pos, quat); joints of a
    body applied in order (slide: translate along the current axis; hinge: rotate about the
    anchor); free joint: pose read from qpos with the quaternion normalised.
    """
    nl = m.nlink
    xpos = np.zeros((nl, 3)); xquat = np.zeros((nl, 4)); xquat[0, 0] = 1
    for l in range(1, nl):
        if m.link_free[l]:
            a = m.link_qposadr[l]
            xpos[l] = qpos[a:a + 3]
            xquat[l] = quat_normalize(qpos[a + 3:a + 7])
            continue
        p = m.link_parent[l]
        R = quat_to_mat(xquat[p])
        pos = xpos[p] + R @ m.link_pos[l]
        quat = quat_mul(xquat[p], m.link_quat[l])
        for d in range(m.link_dofadr[l], m.link_dofadr[l] + m.link_dofnum[l]):
            q = qpos[m.dof_qposadr[d]]
            Rl = quat_to_mat(quat)
            if m.dof_type[d] == DOF_SLIDE:
                pos = pos + Rl @ m.dof_axis[d] * q
            else:
                anchor = pos + Rl @ m.dof_pos[d]
                quat = quat_mul(quat, axis_angle_quat(m.dof_axis[d], q))
                pos = anchor - quat_to_mat(quat) @ m.dof_pos[d]
        xpos[l], xquat[l] = pos, quat_normalize(quat)
    return xpos, xquat


def dof_motion(m: Model, xpos, xquat, qpos):
    """World-frame motion axes: for each dof (ang[3], lin-axis[3], anchor[3])."""
    nv = m.nv
    ang = np.zeros((nv, 3)); lin = np.zeros((nv, 3)); anchor = np.zeros((nv, 3))
    for l in range(1, m.nlink):
        R = quat_to_mat(xquat[l])
        if m.link_free[l]:
            d0 = m.link_dofadr[l]
            for k in range(3):
                lin[d0 + k, k] = 1.0
                ang[d0 + 3 + k] = R[:, k]
                anchor[d0 + 3 + k] = xpos[l]
            continue
        # joints applied in order: axis of joint d is expressed in the frame *after* earlier
        # joints of the same body; for this model a body never mixes hinges (all slides or one
        # hinge), so the final link frame gives the same axes.
        for d in range(m.link_dofadr[l], m.link_dofadr[l] + m.link_dofnum[l]):
            ax = R @ m.dof_axis[d]
            if m.dof_type[d] == DOF_SLIDE:
                lin[d] = ax
            else:
                ang[d] = ax
                anchor[d] = xpos[l] + R @ m.dof_pos[d]
    return ang, lin, anchor


def point_jacobian(m: Model, ang, lin, anchor, link: int, point):
    """jacp[3,nv], jacr[3,nv] of a world point rigidly attached to ``link``."""
    jp = np.zeros((3, m.nv)); jr = np.zeros((3, m.nv))
    if link == 0:
        return jp, jr
    d = m.link_dofadr[link] + m.link_dofnum[link] - 1
    while d >= 0:
        jr[:, d] = ang[d]
        jp[:, d] = lin[d] + np.cross(ang[d], point - anchor[d])
        d = m.dof_parent[d]
    return jp, jr


def mass_matrix(m: Model, qpos):
    xpos, xquat = link_kinematics(m, qpos)
    ang, lin, anchor = dof_motion(m, xpos, xquat, qpos)
    M = np.zeros((m.nv, m.nv))
    for l in range(1, m.nlink):
        R = quat_to_mat(xquat[l])
        c = xpos[l] + R @ m.link_com[l]
        ixx, iyy, izz, ixy, ixz, iyz = m.link_inertia[l]
        I = R @ np.array([[ixx, ixy, ixz], [ixy, iyy, iyz], [ixz, iyz, izz]]) @ R.T
        jp, jr = point_jacobian(m, ang, lin, anchor, l, c)
        M += m.link_mass[l] * jp.T @ jp + jr.T @ I @ jr
    return M


# ----------------------------------------------------------------------------- compile
def compile_model(dofs: Sequence[str] = ("slide_x", "slide_y"), n_blocks: int = 0,
                  block_pos: Optional[np.ndarray] = None, xml_file: str = "models/world.xml",
                  ref_root: Path = DEFAULT_REF_ROOT, set_xml: Sequence = (), block_geom: Optional[dict] = None,
                  plane_convex_points: int = 1) -> Model:
    """plane_convex_points: contacts a plane <-> convex (mesh / cylinder) pair may hold - 1: the deepest support point only (the committed
    reference configurations); 4: up to three more around it, as MuJoCo's mjc_PlaneConvex adds them (oracle/hsr_oracle.c)."""
    ref_root = Path(ref_root)
    assert plane_convex_points in (1, 4)
    if block_pos is None:
        # resting height on the pan: 0.405 + 0.017 (world.xml:83-84, util.py:120)
        block_pos = np.array([[0.0, 0.12 * (i - (n_blocks - 1) / 2.0), 0.422]
                              for i in range(n_blocks)]).reshape(n_blocks, 3)
    block_pos = np.asarray(block_pos, dtype=np.float64).reshape(n_blocks, 3)
    set_xml = [(str(p), str(v)) for p, v in set_xml]
    parsed = _parse_tree(ref_root, xml_file, list(dofs), n_blocks, block_pos, set_xml, block_geom)
    bodies: List[_Body] = parsed["bodies"]
    opt = parsed["opt"]
    meta = dict(parsed["meta"])
    if set_xml:
        meta.update(set_xml=[list(c) for c in set_xml])
    meta.update(plane_convex_points=int(plane_convex_points))
    if block_geom:
        meta.update(block_geom=dict(block_geom))
    meta.update(dofs=list(dofs), n_blocks=n_blocks, xml_file=xml_file,
                decisions="H1 malformed pos->0; H2 inertiafromgeom all geoms density 1000 (legacy "
                          "mesh inertia); H3 default class 'all' is global; H4 hinge ranges in "
                          "degrees; H5 quats normalised; H6 goal is mocap; H7 MuJoCo 2.0 defaults")
    nb = len(bodies)

    # ---- per-mesh data -----------------------------------------------------------------------
    mesh_cache: Dict[str, dict] = {}

    def mesh_data(name):
        if name not in mesh_cache:
            tris = load_stl(parsed["meshes"][name])
            V, com, I = mesh_inertia_legacy(tris)
            hull = convex_hull_vertices(tris.reshape(-1, 3))
            mesh_cache[name] = dict(V=V, com=com, I=I, hull=hull)
        return mesh_cache[name]

    # ---- body inertias in body frame (H2) --------------------------------------------------------
    def geom_inertia(g: _Geom):
        """-> (mass, com in body frame, inertia about com in body frame)"""
        Rg = quat_to_mat(g.quat)
        if g.type == GEOM_PLANE:
            return 0.0, np.zeros(3), np.zeros((3, 3))
        if g.type == GEOM_MESH:
            md = mesh_data(g.mesh)
            vol, c, I = md["V"], md["com"], md["I"]
        elif g.type == GEOM_BOX:
            a, b, c_ = g.size
            vol = 8 * a * b * c_
            I = vol / 3.0 * np.diag([b * b + c_ * c_, a * a + c_ * c_, a * a + b * b]); c = np.zeros(3)
        elif g.type == GEOM_SPHERE:
            r = g.size[0]
            vol = 4.0 / 3.0 * np.pi * r ** 3
            I = 0.4 * vol * r * r * np.eye(3); c = np.zeros(3)
        elif g.type == GEOM_CYLINDER:
            r, hh = g.size[0], g.size[1]
            vol = np.pi * r * r * 2 * hh
            ixx = vol * (3 * r * r + 4 * hh * hh) / 12.0
            I = np.diag([ixx, ixx, 0.5 * vol * r * r]); c = np.zeros(3)
        else:
            raise ValueError(g.type)
        mass = g.mass if g.mass is not None else g.density * vol
        scale = mass / vol
        return mass, g.pos + Rg @ c, Rg @ (I * scale) @ Rg.T

    body_mass = np.zeros(nb); body_com = np.zeros((nb, 3)); body_I = np.zeros((nb, 3, 3))
    for i, b in enumerate(bodies):
        if i == 0:
            continue
        parts = [geom_inertia(g) for g in b.geoms] if parsed["inertiafromgeom"] else []
        parts = [p for p in parts if p[0] > 0]
        if parts:
            mtot = sum(p[0] for p in parts)
            com = sum(p[0] * p[1] for p in parts) / mtot
            I = np.zeros((3, 3))
            for mm, c, Ic in parts:
                d = c - com
                I += Ic + mm * (d @ d * np.eye(3) - np.outer(d, d))
            body_mass[i], body_com[i], body_I[i] = mtot, com, I
        elif b.inertial is not None:
            Ri = quat_to_mat(b.inertial["quat"])
            body_mass[i] = b.inertial["mass"]
            body_com[i] = b.inertial["pos"]
            body_I[i] = Ri @ np.diag(b.inertial["diag"]) @ Ri.T

    # ---- fold joint-less bodies into links ---------------------------------------------------------
    body_link = np.zeros(nb, dtype=np.int32)
    body_lpos = np.zeros((nb, 3)); body_lquat = np.tile([1., 0, 0, 0], (nb, 1))
    links = [dict(body=0, parent=0, pos=np.zeros(3), quat=np.array([1., 0, 0, 0]))]
    for i in range(1, nb):
        b = bodies[i]
        p = b.parent
        # pose of this body in its parent's link frame
        Rp = quat_to_mat(body_lquat[p])
        pos_in_l = body_lpos[p] + Rp @ b.pos
        quat_in_l = quat_normalize(quat_mul(body_lquat[p], b.quat))
        if b.joints:
            links.append(dict(body=i, parent=int(body_link[p]), pos=pos_in_l, quat=quat_in_l))
            body_link[i] = len(links) - 1
        else:
            body_link[i] = body_link[p]
            body_lpos[i], body_lquat[i] = pos_in_l, quat_in_l
    nlink = len(links)

    link_mass = np.zeros(nlink); link_com = np.zeros((nlink, 3)); link_I = np.zeros((nlink, 3, 3))
    for l in range(1, nlink):
        members = [i for i in range(nb) if body_link[i] == l and body_mass[i] > 0]
        mtot = sum(body_mass[i] for i in members)
        coms = {i: body_lpos[i] + quat_to_mat(body_lquat[i]) @ body_com[i] for i in members}
        com = sum(body_mass[i] * coms[i] for i in members) / mtot
        I = np.zeros((3, 3))
        for i in members:
            Rb = quat_to_mat(body_lquat[i])
            d = coms[i] - com
            I += Rb @ body_I[i] @ Rb.T + body_mass[i] * (d @ d * np.eye(3) - np.outer(d, d))
        link_mass[l], link_com[l], link_I[l] = mtot, com, I

    # ---- dofs -----------------------------------------------------------------------------------------
    dof = dict(link=[], type=[], axis=[], pos=[], parent=[], damping=[], qposadr=[], limited=[],
               range=[])
    link_dofadr = np.zeros(nlink, dtype=np.int32); link_dofnum = np.zeros(nlink, dtype=np.int32)
    link_qposadr = np.zeros(nlink, dtype=np.int32); link_free = np.zeros(nlink, dtype=np.int32)
    joint_names, joint_qposadr, joint_dofadr = [], [], []
    qpos0 = []
    last_dof_of_link = {0: -1}
    for l in range(1, nlink):
        b = bodies[links[l]["body"]]
        link_dofadr[l] = len(dof["link"]); link_qposadr[l] = len(qpos0)
        prev = last_dof_of_link[links[l]["parent"]]
        for j in b.joints:
            joint_names.append(j.name)
            if j.type == "free":
                assert len(b.joints) == 1 and links[l]["parent"] == 0
                link_free[l] = 1
                joint_qposadr.append((len(qpos0), 7)); joint_dofadr.append(len(dof["link"]))
                qa = len(qpos0)
                qpos0 += list(links[l]["pos"]) + list(links[l]["quat"])
                for k in range(6):
                    dof["link"].append(l); dof["type"].append(DOF_FREE_LIN if k < 3 else DOF_FREE_ANG)
                    dof["axis"].append(np.eye(3)[k % 3]); dof["pos"].append(np.zeros(3))
                    dof["parent"].append(prev); prev = len(dof["link"]) - 1
                    # translational dofs address qpos[qa+k]; rotational ones the quaternion start
                    dof["damping"].append(0.0); dof["qposadr"].append(qa + k if k < 3 else qa + 3)
                    dof["limited"].append(0); dof["range"].append(np.zeros(2))
            else:
                joint_qposadr.append((len(qpos0), 1)); joint_dofadr.append(len(dof["link"]))
                dof["link"].append(l); dof["type"].append(DOF_SLIDE if j.type == "slide" else DOF_HINGE)
                dof["axis"].append(j.axis); dof["pos"].append(j.pos); dof["parent"].append(prev)
                prev = len(dof["link"]) - 1
                dof["damping"].append(j.damping); dof["qposadr"].append(len(qpos0))
                dof["limited"].append(int(j.limited)); dof["range"].append(j.range)
                qpos0.append(0.0)
        link_dofnum[l] = len(dof["link"]) - link_dofadr[l]
        last_dof_of_link[l] = prev
    nv, nq = len(dof["link"]), len(qpos0)
    # a non-free link may hold several slides or exactly one hinge (keeps dof axes = final frame)
    for l in range(1, nlink):
        types = [dof["type"][d] for d in range(link_dofadr[l], link_dofadr[l] + link_dofnum[l])]
        assert link_free[l] or types.count(DOF_HINGE) == 0 or len(types) == 1, "mixed joints on a body"

    # ---- geoms (collidable only) -------------------------------------------------------------------------
    g_rec = []
    mesh_vert = []
    geom_names = []
    for i, b in enumerate(bodies):
        for g in b.geoms:
            if g.contype == 0 and g.conaffinity == 0:
                continue
            Rb = quat_to_mat(body_lquat[i])
            pos = body_lpos[i] + Rb @ g.pos
            quat = quat_normalize(quat_mul(body_lquat[i], g.quat))
            meshadr, meshnum = 0, 0
            if g.type == GEOM_MESH:
                md = mesh_data(g.mesh)
                # geom frame origin := mesh centre of mass (MuJoCo recentres meshes)
                verts = md["hull"] - md["com"]
                pos = pos + quat_to_mat(quat) @ md["com"]
                meshadr, meshnum = sum(len(v) for v in mesh_vert), len(verts)
                mesh_vert.append(verts)
                rbound = np.linalg.norm(verts, axis=1).max()
                size = np.abs(verts).max(0)
            elif g.type == GEOM_BOX:
                rbound = np.linalg.norm(g.size); size = g.size
            elif g.type == GEOM_CYLINDER:
                rbound = np.hypot(g.size[0], g.size[1]); size = g.size
            elif g.type == GEOM_SPHERE:
                rbound = g.size[0]; size = g.size
            else:
                rbound = 0.0; size = g.size
            # box containing the geom in its own frame: centre(3) + half extents(3); tight (asymmetric) for hulls
            if g.type == GEOM_MESH:
                lo_, hi_ = verts.min(0), verts.max(0)
                aabb = np.concatenate([(lo_ + hi_) / 2, (hi_ - lo_) / 2])
            elif g.type == GEOM_CYLINDER:
                aabb = np.array([0, 0, 0, g.size[0], g.size[0], g.size[1]])
            elif g.type == GEOM_SPHERE:
                aabb = np.array([0, 0, 0, g.size[0], g.size[0], g.size[0]])
            elif g.type == GEOM_BOX:
                aabb = np.concatenate([np.zeros(3), g.size])
            else:
                aabb = np.zeros(6)
            g_rec.append(dict(type=g.type, link=int(body_link[i]), body=i, pos=pos, quat=quat, aabb=aabb,
                              size=size, rbound=rbound, condim=g.condim, meshadr=meshadr,
                              meshnum=meshnum, g=g))
            geom_names.append(g.name if g.mesh is None else f"{b.name}:{g.mesh}")
    ngeom = len(g_rec)
    mesh_vert = np.concatenate(mesh_vert) if mesh_vert else np.zeros((0, 3))

    # ---- candidate pairs ---------------------------------------------------------------------------------------
    link_parent = np.array([lk["parent"] for lk in links], dtype=np.int32)
    excl = {frozenset(e) for e in parsed["excludes"]}
    pairs = []
    for a in range(ngeom):
        for c in range(a + 1, ngeom):
            ga, gc = g_rec[a], g_rec[c]
            la, lc = ga["link"], gc["link"]
            if la == lc:
                continue
            if (link_parent[la] == lc and lc != 0) or (link_parent[lc] == la and la != 0):
                continue
            if frozenset((bodies[ga["body"]].name, bodies[gc["body"]].name)) in excl:
                continue
            if not ((ga["g"].contype & gc["g"].conaffinity) or (gc["g"].contype & ga["g"].conaffinity)):
                continue
            g1, g2 = (a, c) if ga["type"] <= gc["type"] else (c, a)
            t1, t2 = g_rec[g1]["type"], g_rec[g2]["type"]
            if t1 == GEOM_PLANE:
                fn = FN_PLANE_BOX if t2 == GEOM_BOX else FN_PLANE_CONVEX
            elif t1 == GEOM_BOX and t2 == GEOM_BOX:
                fn = FN_BOX_BOX
            else:
                fn = FN_CONVEX
            A, B = g_rec[g1]["g"], g_rec[g2]["g"]
            fr = np.maximum(A.friction, B.friction)
            pairs.append(dict(g1=g1, g2=g2, fn=fn, condim=max(A.condim, B.condim),
                              friction=np.array([fr[0], fr[0], fr[1], fr[2], fr[2]]),
                              solref=0.5 * (A.solref + B.solref), solimp=0.5 * (A.solimp + B.solimp)))
    npair = len(pairs)
    slot = np.zeros(npair + 1, dtype=np.int32)
    for i, p in enumerate(pairs):
        slot[i + 1] = slot[i] + (plane_convex_points if p["fn"] == FN_PLANE_CONVEX else FN_MAXCON[p["fn"]])

    # ---- actuators -------------------------------------------------------------------------------------------------
    acts = parsed["actuators"]
    act_dof = np.array([joint_dofadr[joint_names.index(a.get("joint"))] for a in acts], dtype=np.int32)
    act_gear = np.array([float(a.get("gear", 1)) for a in acts])
    act_kp = np.array([float(a.get("kp", 1)) for a in acts])
    # MuJoCo clamps ctrl / actuator force only when ctrllimited / forcelimited is set (world.xml:104-124 sets both on all seven
    # actuators); an unlimited actuator gets an effectively infinite range so that the kernels' unconditional clamp is a no-op
    def _range(a, key, flag):
        if a.get(flag, "false") != "true" or a.get(key) is None:
            return [-UNLIMITED, UNLIMITED]
        return _vec(a.get(key), 2, [0, 0])
    act_ctrlrange = np.array([_range(a, "ctrlrange", "ctrllimited") for a in acts], dtype=np.float64).reshape(-1, 2)
    act_forcerange = np.array([_range(a, "forcerange", "forcelimited") for a in acts], dtype=np.float64).reshape(-1, 2)

    # bit k of link_dofmask[l] is set when dof k lies on the path from link l to the root
    link_dofmask = np.zeros(nlink, dtype=np.int32)
    for l in range(1, nlink):
        k = link_dofadr[l] + link_dofnum[l] - 1
        while k >= 0:
            link_dofmask[l] |= (1 << k)
            k = dof["parent"][k]
    # device-friendly buffer caps (the XML asks for nconmax=100 njmax=500, world.xml:44, which MuJoCo only
    # uses as buffer sizes): contacts beyond nconmax / rows beyond njmax are dropped identically by the
    # oracle and the HIP path.  One lane group (16 or 32 lanes) serves an env, so caps are multiples of it.
    group = 16 if nv <= 16 else 32
    eff_nconmax = group
    # 32-lane models: 124 rows - measured on cfg4 (8192 envs x 300 env-steps of the bench: 7.4e8 env-substeps) 3.1e5 substeps wanted 97-104 rows,
    # 8.9e3 105-112, 1.4e3 113-120, 24 121-128, 1 more: with 124 fewer than 4e-8 of the env-substeps drop a row (96 rows: 4.3e-4); 124 is what
    # the 20 KB of LDS per workgroup hold once the pair / geom tables are read from global memory (persist.h: TG)
    eff_njmax = {True: 48, False: 124}[nv <= 16]
    meta["xml_nconmax_njmax"] = [opt["nconmax"], opt["njmax"]]
    opt["nconmax"], opt["njmax"] = eff_nconmax, eff_njmax

    arrays = dict(
        qpos0=np.array(qpos0),
        link_dofmask=link_dofmask,
        link_parent=link_parent,
        link_pos=np.array([lk["pos"] for lk in links]), link_quat=np.array([lk["quat"] for lk in links]),
        link_dofadr=link_dofadr, link_dofnum=link_dofnum, link_qposadr=link_qposadr, link_free=link_free,
        link_mass=link_mass, link_com=link_com,
        link_inertia=np.array([[I[0, 0], I[1, 1], I[2, 2], I[0, 1], I[0, 2], I[1, 2]] for I in link_I]),
        dof_link=np.array(dof["link"], dtype=np.int32), dof_type=np.array(dof["type"], dtype=np.int32),
        dof_axis=np.array(dof["axis"]).reshape(nv, 3), dof_pos=np.array(dof["pos"]).reshape(nv, 3),
        dof_parent=np.array(dof["parent"], dtype=np.int32), dof_damping=np.array(dof["damping"]),
        dof_qposadr=np.array(dof["qposadr"], dtype=np.int32),
        dof_invweight0=np.zeros(nv), dof_limited=np.array(dof["limited"], dtype=np.int32),
        dof_range=np.array(dof["range"]).reshape(nv, 2),
        dof_solref=np.tile([0.02, 1.0], (nv, 1)), dof_solimp=np.tile([0.9, 0.95, 0.001, 0.5, 2.0], (nv, 1)),
        body_link=body_link, body_pos=body_lpos, body_quat=body_lquat,
        body_mocap=np.array([int(b.mocap) for b in bodies], dtype=np.int32),
        geom_type=np.array([g["type"] for g in g_rec], dtype=np.int32),
        geom_link=np.array([g["link"] for g in g_rec], dtype=np.int32),
        geom_body=np.array([g["body"] for g in g_rec], dtype=np.int32),
        geom_pos=np.array([g["pos"] for g in g_rec]), geom_quat=np.array([g["quat"] for g in g_rec]),
        geom_size=np.array([g["size"] for g in g_rec]), geom_rbound=np.array([g["rbound"] for g in g_rec]),
        geom_condim=np.array([g["condim"] for g in g_rec], dtype=np.int32),
        geom_meshadr=np.array([g["meshadr"] for g in g_rec], dtype=np.int32),
        geom_meshnum=np.array([g["meshnum"] for g in g_rec], dtype=np.int32),
        geom_invweight=np.zeros((ngeom, 2)),
        geom_aabb=np.array([g["aabb"] for g in g_rec]).reshape(ngeom, 6),
        mesh_vert=mesh_vert,
        pair_geom1=np.array([p["g1"] for p in pairs], dtype=np.int32),
        pair_geom2=np.array([p["g2"] for p in pairs], dtype=np.int32),
        pair_fn=np.array([p["fn"] for p in pairs], dtype=np.int32),
        pair_condim=np.array([p["condim"] for p in pairs], dtype=np.int32),
        pair_slot=slot,
        pair_friction=np.array([p["friction"] for p in pairs]).reshape(npair, 5),
        pair_solref=np.array([p["solref"] for p in pairs]).reshape(npair, 2),
        pair_solimp=np.array([p["solimp"] for p in pairs]).reshape(npair, 5),
        act_dof=act_dof, act_gear=act_gear, act_kp=act_kp, act_ctrlrange=act_ctrlrange,
        act_forcerange=act_forcerange,
    )
    sizes = np.zeros(16, dtype=np.int32)
    sizes[[SZ_NQ, SZ_NV, SZ_NU, SZ_NLINK, SZ_NBODY, SZ_NGEOM, SZ_NPAIR, SZ_NMESHVERT, SZ_NSLOT,
           SZ_NLIMIT, SZ_NCONMAX, SZ_NJMAX, SZ_NMOCAP]] = [
        nq, nv, len(acts), nlink, nb, ngeom, npair, len(mesh_vert), slot[-1],
        int(np.sum(dof["limited"])), opt["nconmax"], opt["njmax"], sum(b.mocap for b in bodies)]
    optv = np.zeros(16)
    optv[[OPT_TIMESTEP, OPT_IMPRATIO, OPT_GRAV_Z, OPT_TOLERANCE, OPT_ITERATIONS, OPT_LS_ITERATIONS,
          OPT_LS_TOLERANCE, OPT_MPR_TOLERANCE, OPT_MPR_ITERATIONS]] = [
        opt["timestep"], opt["impratio"], -9.81, 1e-8, 100, 50, 0.01, 1e-6, 50]
    arrays["sizes"], arrays["opt"] = sizes, optv
    assert opt["cone"] == "elliptic", "only the reference's elliptic cones are implemented"

    names = dict(body=[b.name for b in bodies], joint=joint_names, geom=geom_names,
                 actuator=[a.get("name") for a in acts],
                 link=[bodies[lk["body"]].name for lk in links])
    meta["joint_qposadr"] = joint_qposadr
    meta["joint_dofadr"] = joint_dofadr
    meta["mesh_license"] = ("hull vertices derived from hsr/hsr_meshes (Toyota, CC BY-NC-ND 4.0, "
                            "hsr/hsr_meshes/LICENSE.txt); kept only as collision tables")
    model = Model(arrays=arrays, names=names, meta=meta)

    # ---- qpos0 statistics (mj_setConst) ------------------------------------------------------------------------------
    q0 = arrays["qpos0"]
    M0 = mass_matrix(model, q0)
    Minv = np.linalg.inv(M0)
    optv[OPT_MEANINERTIA] = np.trace(M0) / nv
    dinv = np.diag(Minv).copy()
    for l in range(1, nlink):
        if link_free[l]:
            a = link_dofadr[l]
            dinv[a:a + 3] = dinv[a:a + 3].mean(); dinv[a + 3:a + 6] = dinv[a + 3:a + 6].mean()
    arrays["dof_invweight0"][:] = dinv
    # dofs >= ndense never couple to another dof in M (e.g. a free box with its COM at the body origin):
    # the cooperative Cholesky skips their off-diagonal updates when factoring M and M + h B
    rs = np.random.default_rng(0)
    dense = 0
    for _ in range(4):
        qr = q0.copy()
        qr[:len(acts)] += rs.uniform(-0.3, 0.3, len(acts))
        for l in range(1, nlink):
            if link_free[l]:
                a = link_qposadr[l]
                qr[a + 3:a + 7] = quat_normalize(rs.normal(size=4))
        Mr = mass_matrix(model, qr)
        off = np.abs(Mr - np.diag(np.diag(Mr))) > 1e-12
        idx = np.where(off.any(axis=0))[0]
        dense = max(dense, int(idx.max()) + 1 if idx.size else 0)
    sizes[SZ_NDENSE] = dense
    xpos, xquat = link_kinematics(model, q0)
    ang, lin, anchor = dof_motion(model, xpos, xquat, q0)
    body_invw = np.zeros((nb, 2))
    for i in range(1, nb):
        l = body_link[i]
        if l == 0:
            continue
        Rl = quat_to_mat(xquat[l])
        c = xpos[l] + Rl @ (body_lpos[i] + quat_to_mat(body_lquat[i]) @ body_com[i])
        jp, jr = point_jacobian(model, ang, lin, anchor, l, c)
        body_invw[i, 0] = np.trace(jp @ Minv @ jp.T) / 3.0
        body_invw[i, 1] = np.trace(jr @ Minv @ jr.T) / 3.0
    for k, g in enumerate(g_rec):
        arrays["geom_invweight"][k] = body_invw[g["body"]]
    meta["body_invweight0"] = body_invw.tolist()
    meta["link_mass"] = link_mass.tolist()
    return model


# the four benchmark configurations of BASELINE.json
CONFIGS = {
    "cfg1": dict(dofs=["slide_x", "slide_y"], n_blocks=0),
    "cfg2": dict(dofs=["slide_x", "slide_y"], n_blocks=1),
    "cfg3": dict(dofs=ALL_DOFS, n_blocks=1),
    "cfg4": dict(dofs=ALL_DOFS, n_blocks=3),
    # the cupboard scene (cupboard-world.xml:91-116) with its own `block` / `blockjoint`, the shape of the
    # hsr/__init__.py:10-19 demo; many more box geoms -> 274 candidate pairs
    "cupboard": dict(dofs=ALL_DOFS, n_blocks=0, xml_file="models/cupboard-world.xml"),
}
# further committed blobs, used by the parity tests only (the GPU box has no reference tree to compile from):
#   cfg3_setxml  SURVEY 8f row 2: cfg3 through `--set-xml` (pan friction halved, arm-lift actuator without ctrl limit)
#   nq18         a model whose qpos (18) does not fit the 16 lanes its 16 dofs select: must fall back to the per-substep chain
TEST_CONFIGS = {
    "cfg3_setxml": dict(dofs=ALL_DOFS, n_blocks=1,
                        set_xml=[("worldbody/body[@name='pan']/geom/friction", "0.5 0.005 0.0001"),
                                 ("actuator/position[@name='arm_lift_motor']/ctrllimited", "false")]),
    "nq18": dict(dofs=["slide_x", "slide_y", "arm_lift_joint", "arm_flex_joint"], n_blocks=2),
    # sizes none of the reference configurations has: they run the GENERIC instances of the persistent kernel (nv and ndense at
    # run time) - nv 11 in a 16-lane group, nv 23 in a 32-lane group
    "nv11": dict(dofs=["slide_x", "slide_y", "arm_lift_joint", "arm_flex_joint", "wrist_roll_joint"], n_blocks=1),
    "nv23": dict(dofs=["slide_x", "slide_y", "arm_lift_joint", "arm_flex_joint", "wrist_roll_joint"], n_blocks=3),
    # no robot dof at all: the robot is scenery (15 static hulls + the wrist cylinder), the block the only body - thrown around it, it
    # leaves and re-enters the cull ranges of the hulls (the separation-margin stamps of the convex pairs: tests/test_gpu_hotpath.py)
    "static1": dict(dofs=[], n_blocks=1),
    # plane <-> convex with several points (round 4): the head-pan hull (32 vertices, a flat bottom face of 61 cm^2) as a free body lying on
    # the FLOOR plane, away from the robot; the same scene with the single deepest point for comparison (tests/test_kat.py)
    "meshrest4": dict(dofs=[], n_blocks=1, block_geom=dict(type="mesh", mesh="head_pan"), plane_convex_points=4,
                      block_pos=np.array([[1.0, 0.6, 0.05]])),
    "meshrest1": dict(dofs=[], n_blocks=1, block_geom=dict(type="mesh", mesh="head_pan"), plane_convex_points=1,
                      block_pos=np.array([[1.0, 0.6, 0.05]])),
}
MODEL_DIR = Path(__file__).parent / "models"


def load_config(name: str) -> Model:
    """Load a committed blob (no reference tree needed)."""
    return Model.load(MODEL_DIR / f"{name}.hsrm")


def emit_mjcf(outdir, name: str, dofs: Sequence[str] = ("slide_x", "slide_y"), n_blocks: int = 0,
              block_pos: Optional[np.ndarray] = None, xml_file: str = "models/world.xml",
              ref_root: Path = DEFAULT_REF_ROOT, set_xml: Sequence = ()) -> Path:
    """Write the MJCF that the reference's launcher would hand to mujoco-py for this configuration - the mutations of
    hsr/util.py:93-159 (block injection, --set-xml, actuator / joint filter, include and meshdir paths) applied to the
    main file and to every included file - as <outdir>/<name>.xml (+ <name>__<include>).  Anyone with a MuJoCo install
    can load it to generate external golden vectors (tests/test_mujoco_crosscheck.py)."""
    ref_root, outdir = Path(ref_root), Path(outdir)
    outdir.mkdir(parents=True, exist_ok=True)
    xml_path = ref_root / xml_file
    if block_pos is None:
        block_pos = np.array([[0.0, 0.12 * (i - (n_blocks - 1) / 2.0), 0.422] for i in range(n_blocks)]).reshape(n_blocks, 3)
    set_xml = [(str(p), str(v)) for p, v in set_xml]
    includes = [e.get("file") for e in ET.parse(xml_path).findall("*/include")]
    out_main = outdir / f"{name}.xml"
    for rel in [None] + includes:
        src = xml_path if rel is None else xml_path.parent / rel
        tree = ET.parse(src)
        root = tree.getroot()
        worldbody = root.find("./worldbody")
        if worldbody is not None:
            for i in range(n_blocks):
                body = ET.SubElement(worldbody, "body", attrib=dict(name=f"block{i}", pos=" ".join(repr(float(x)) for x in block_pos[i])))
                ET.SubElement(body, "geom", attrib=dict(name=f"block{i}", type="box", mass="1", size=".05 .025 .017", condim="6",
                                                        solimp="0.99 0.99 0.01", solref="0.01 1"))
                ET.SubElement(body, "freejoint", attrib=dict(name=f"block{i}joint"))
        _apply_setters(root, set_xml)
        for acts in root.iter("actuator"):
            for a in list(acts):
                if a.get("joint") not in dofs:
                    acts.remove(a)
        for body in root.iter("body"):
            for j in body.findall("joint"):
                if j.get("name") not in dofs:
                    body.remove(j)
        for inc in root.findall("*/include"):
            inc.set("file", f"{name}__{Path(inc.get('file')).name}")
        for comp in root.findall("compiler"):
            comp.set("meshdir", str((xml_path.parent / comp.get("meshdir", ".")).resolve()))
        tree.write(out_main if rel is None else outdir / f"{name}__{Path(rel).name}")
    return out_main


def main():
    import argparse
    ap = argparse.ArgumentParser(description="compile the committed model blobs / emit the mutated MJCF of each configuration")
    ap.add_argument("--emit-mjcf", metavar="DIR", default=None,
                    help="write <DIR>/<config>.xml (what hsr/util.py:mutate_xml yields) for every configuration instead of compiling")
    args = ap.parse_args()
    every = dict(CONFIGS, **TEST_CONFIGS)
    if args.emit_mjcf:
        for name, kw in every.items():
            print(emit_mjcf(args.emit_mjcf, name, **kw))
        return
    MODEL_DIR.mkdir(exist_ok=True)
    for name, kw in every.items():
        m = compile_model(**kw)
        m.save(MODEL_DIR / f"{name}.hsrm")
        print(name, "nq", m.nq, "nv", m.nv, "nu", m.nu, "nlink", m.nlink, "nbody", m.nbody,
              "ngeom", m.ngeom, "npair", m.npair, "nslot", m.nslot, "bytes", len(m.to_bytes()))


if __name__ == "__main__":
    main()
