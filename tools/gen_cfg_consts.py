"""Generate hsr_env_amd/csrc/cfg_consts.h from the committed model blobs: the scalar fields of DevModel (sizes, solver options) of every
reference configuration as static constexpr members of a type derived from DevModel, which the persistent kernel takes as its
template parameter MT (persist.h) - `m.nlink` is then a literal and the LDS layout a set of immediates.  hsr_batch_create compares the
table at the end of the header with the loaded model and falls back to the run-time instances when anything differs.

    python tools/gen_cfg_consts.py            # rewrite the header
    python tools/gen_cfg_consts.py --check    # exit 1 if the committed header is stale (tests/test_compiler.py)"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from hsr_env_amd.compiler import load_config  # noqa: E402

CONFIGS = ["cfg1", "cfg2", "cfg3", "cfg4", "cupboard"]
INTS = ["nq", "nv", "nu", "nlink", "nbody", "ngeom", "npair", "nslot", "nconmax", "njmax", "nM", "ndense", "npair_pad", "nstatic_geom", "nfb",
        "maxdepth", "iterations", "ls_iterations", "mpr_iterations", "any_damping", "solimp_general"]
FLOATS = ["timestep", "impratio", "gravz", "tolerance", "ls_tolerance", "mpr_tolerance", "meaninertia"]


def consts(m):
    a = m.arrays
    sz, op = a["sizes"], a["opt"]
    nq, nv, nu, nlink, nbody, ngeom, npair, _, nslot, _, nconmax, njmax, _, ndense = [int(x) for x in sz[:14]]
    lp = a["link_parent"]
    dep = [0] * nlink
    for l in range(1, nlink):
        dep[l] = dep[int(lp[l])] + 1
    gl = a["geom_link"]
    nstat = 0
    while nstat < ngeom and gl[nstat] == 0:
        nstat += 1
    # trailing free bodies (hsrsim.hip batch_init): link l owns exactly the dofs [nv - 6 (k + 1), nv - 6 k), lin then ang
    dn, lf, da, dt, dl = a["link_dofnum"], a["link_free"], a["link_dofadr"], a["dof_type"], a["dof_link"]
    group = 16 if nv <= 16 else 32
    nfb = 0
    k = 0
    while 6 * (k + 1) <= nv:
        a0 = nv - 6 * (k + 1)
        l = int(dl[a0])
        fb = l > 0 and lf[l] and da[l] == a0 and dn[l] == 6 and all(dl[a0 + j] == l and dt[a0 + j] == (2 if j < 3 else 3) for j in range(6))
        if not fb:
            break
        nfb += 1
        k += 1
    if nfb * 28 * (64 // group) > 4 * 48:
        nfb = 0
    ints = dict(nq=nq, nv=nv, nu=nu, nlink=nlink, nbody=nbody, ngeom=ngeom, npair=npair, nslot=nslot, nconmax=nconmax, njmax=njmax,
                nM=nv * (nv + 1) // 2, ndense=ndense, npair_pad=max((npair + 7) & ~7, 8), nstatic_geom=nstat, nfb=nfb, maxdepth=max(dep),
                iterations=int(op[4]), ls_iterations=int(op[5]), mpr_iterations=int(op[8]), any_damping=int((a["dof_damping"] > 0).any()),
                solimp_general=int(any(float(max(p, 1.0)) not in (1.0, 2.0) for arr in (a["dof_solimp"], a["pair_solimp"]) for p in arr.reshape(-1, 5)[:, 4])))
    floats = dict(timestep=op[0], impratio=op[1], gravz=op[2], tolerance=op[3], ls_tolerance=op[6], mpr_tolerance=op[7], meaninertia=op[9])
    return ints, floats


def _f(x):
    return f"(float){float(x)!r}"


def quat2mat(q):
    """hsrsim.hip quat2mat_h, operation for operation (double arithmetic, then one rounding to float)"""
    import math
    n = math.sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3])
    w, x, y, z = q[0] / n, q[1] / n, q[2] / n, q[3] / n
    return [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
            2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]


def kin3_tables(cfg, m):
    """The kinematic tree as compile-time tables (csrc/kin3.h) - or None when the model is outside what kin3.h covers: scalar-joint links whose
    dofs are one contiguous run (the robot), free bodies hanging off the world with their com at the origin, principal-axis inertia and identical
    mass properties.  Returns (struct text, check-table ints, check-table floats)."""
    import numpy as np
    a = m.arrays
    sz = a["sizes"]
    nv, nlink = int(sz[1]), int(sz[3])
    lp, lf, da, dn, lq = a["link_parent"], a["link_free"], a["link_dofadr"], a["link_dofnum"], a["link_qposadr"]
    dt, dq, dl = a["dof_type"], a["dof_qposadr"], a["dof_link"]
    robot_links = [l for l in range(1, nlink) if not lf[l]]
    free_links = [l for l in range(1, nlink) if lf[l]]
    if any(lp[l] >= l for l in range(1, nlink)) or any(dn[l] > 3 or dn[l] < 1 for l in robot_links):
        return None
    rdofs = [k for k in range(nv) if not lf[dl[k]]]
    if rdofs and rdofs != list(range(rdofs[0], rdofs[0] + len(rdofs))):
        return None
    if len(rdofs) > 8 or not robot_links:
        return None
    for l in free_links:
        if lp[l] != 0 or dn[l] != 6 or np.abs(a["link_com"][l]).max() != 0 or np.abs(a["link_inertia"][l][3:]).max() != 0:
            return None
        if (a["link_inertia"][l] != a["link_inertia"][free_links[0]]).any() or a["link_mass"][l] != a["link_mass"][free_links[0]]:
            return None
        if [int(dt[da[l] + j]) for j in range(6)] != [2, 2, 2, 3, 3, 3]:
            return None
    lmat = [quat2mat([float(x) for x in a["link_quat"][l]]) for l in range(nlink)]
    nm = "Kin3_" + cfg
    ints = dict(parent=[int(x) for x in lp], isfree=[int(x) for x in lf], dofadr=[int(x) for x in da], dofnum=[int(x) for x in dn], qadr=[int(x) for x in lq],
                dtype=[int(x) for x in dt], dqadr=[int(x) for x in dq], dlink=[int(x) for x in dl])
    fl = dict(lpos=a["link_pos"].reshape(nlink, 3), lmat=np.array(lmat).reshape(nlink, 9), lcom=a["link_com"].reshape(nlink, 3), linr=a["link_inertia"].reshape(nlink, 6),
              lmass=a["link_mass"].reshape(nlink, 1), daxis=a["dof_axis"].reshape(nv, 3), dpos=a["dof_pos"].reshape(nv, 3))
    out = [f"struct {nm} {{", "    static constexpr bool ok = true;",
           f"    static constexpr int NL = {nlink}, NV = {nv}, RD0 = {rdofs[0]}, NRD = {len(rdofs)}, NFREE = {len(free_links)}, F0 = {free_links[0] if free_links else 0};"]
    for k, v in ints.items():
        out.append(f"    static constexpr int {k}[{len(v)}] = {{" + ", ".join(str(x) for x in v) + "};")
    for k, v in fl.items():
        if k == "lmass":
            out.append(f"    static constexpr float lmass[{nlink}] = {{" + ", ".join(_f(x) for x in v[:, 0]) + "};")
        else:
            out.append(f"    static constexpr float {k}[{v.shape[0]}][{v.shape[1]}] = {{" + ", ".join("{" + ", ".join(_f(x) for x in row) + "}" for row in v) + "};")
    out.append("};")
    chk_i = [nlink, nv] + [x for v in ints.values() for x in v]
    chk_f = [float(x) for v in fl.values() for x in np.asarray(v).reshape(-1)]
    out.append(f"static const int kKin3I_{cfg}[] = {{" + ", ".join(str(x) for x in chk_i) + "};")
    out.append(f"static const float kKin3F_{cfg}[] = {{" + ", ".join(_f(x) for x in chk_f) + "};")
    return "\n".join(out)


def render():
    out = ["// GENERATED by tools/gen_cfg_consts.py from hsr_env_amd/models/*.hsrm - do not edit.",
           "// Scalar fields of DevModel as compile-time constants of the reference configurations (persist.h: template parameter MT).",
           "#pragma once", '#include "model.h"', ""]
    rows = []
    out.append("// the kinematic tree of a configuration as compile-time tables (kin3.h); Kin3_none: the tree is read from memory (kin2.h)")
    out.append("struct Kin3_none { static constexpr bool ok = false; };")
    kin3 = {}
    for cfg in CONFIGS:
        m_ = load_config(cfg)
        ints, floats = consts(m_)
        kin3[cfg] = kin3_tables(cfg, m_)
        if kin3[cfg]:
            out.append(kin3[cfg])
        name = "DevModel_" + cfg
        out.append(f"struct {name} : DevModel {{")
        out.append(f"    using kin3 = {'Kin3_' + cfg if kin3[cfg] else 'Kin3_none'};")
        out.append("    static constexpr int " + ", ".join(f"{k} = {ints[k]}" for k in INTS) + ";")
        out.append("    static constexpr float " + ", ".join(f"{k} = (float){float(floats[k])!r}" for k in FLOATS) + ";")
        out.append("};")
        rows.append((cfg, ints, floats))
    out.append("")
    out.append("// host-side copy of the same values: hsr_batch_create picks a constant instance only for a model that matches one row exactly")
    out.append(f"struct CfgConstRow {{ const char *name; int i[{len(INTS)}]; float f[{len(FLOATS)}]; }};")
    out.append("static const CfgConstRow kCfgConsts[] = {")
    for cfg, ints, floats in rows:
        out.append(f'    {{"{cfg}", {{' + ", ".join(str(ints[k]) for k in INTS) + "}, {" + ", ".join(f"(float){float(floats[k])!r}" for k in FLOATS) + "}},")
    out.append("};")
    out.append("// per row of kCfgConsts: the tree tables its instance was compiled with (NULL: none) - hsr_batch_create compares them with the loaded model too")
    out.append("struct Kin3Check { const int *i; int ni; const float *f; int nf; };")
    out.append("static const Kin3Check kKin3Checks[] = {" + ", ".join((f"{{kKin3I_{c}, (int)(sizeof kKin3I_{c} / sizeof(int)), kKin3F_{c}, (int)(sizeof kKin3F_{c} / sizeof(float))}}" if kin3[c] else "{nullptr, 0, nullptr, 0}") for c in CONFIGS) + "};")
    out.append("// order of CfgConstRow::i / ::f: " + " ".join(INTS) + " | " + " ".join(FLOATS))
    out.append("static inline void cfg_const_values(const DevModel &d, int *i, float *f) {")
    out.append("    const int iv[] = {" + ", ".join("d." + k for k in INTS) + "};")
    out.append("    const float fv[] = {" + ", ".join("d." + k for k in FLOATS) + "};")
    out.append(f"    for (int k = 0; k < {len(INTS)}; k++) i[k] = iv[k];")
    out.append(f"    for (int k = 0; k < {len(FLOATS)}; k++) f[k] = fv[k];")
    out.append("}")
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    path = ROOT / "hsr_env_amd" / "csrc" / "cfg_consts.h"
    text = render()
    if "--check" in sys.argv:
        sys.exit(0 if path.exists() and path.read_text() == text else 1)
    path.write_text(text)
    print(path)
