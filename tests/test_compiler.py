"""Model compiler: committed blobs reproduce SURVEY.md section 8's size table and survive (de)serialisation."""
from pathlib import Path

import numpy as np
import pytest

from hsr_env_amd import compiler as hc

# cfg: nq nv nu obs nbody collidable-geoms candidate-pairs   
TABLE = {"cfg1": (2, 2, 2, 4, 47, 17, 30), "cfg2": (9, 8, 2, 17, 48, 18, 47),
         "cfg3": (14, 13, 7, 27, 48, 18, 114), "cfg4": (28, 25, 7, 53, 50, 20, 151),
         "cupboard": (14, 13, 7, 27, 53, 28, 274)}


@pytest.mark.parametrize("cfg", list(TABLE))
def test_sizes_match_survey_table(models, cfg):
    m = models[cfg]
    nq, nv, nu, nobs, nbody, ngeom, npair = TABLE[cfg]
    assert (m.nq, m.nv, m.nu, m.nq + m.nv, m.nbody, m.ngeom, m.npair) == (nq, nv, nu, nobs, nbody, ngeom, npair)
    assert m.timestep == 0.002 and m.opt[hc.OPT_IMPRATIO] == 2.5       # world.xml:2


def test_blob_roundtrip(models):
    m = models["cfg3"]
    m2 = hc.Model.from_bytes(m.to_bytes())
    for k, a in m.arrays.items():
        assert np.array_equal(a, m2.arrays[k]), k
    assert m2.names == m.names and m2.body_id("block0") == m.body_id("block0")
    assert m.joint_qpos_addr("slide_x") == 0 and m.joint_qpos_addr("block0joint") == (7, 14)


def test_model_constants(models):
    """Appendix A of SURVEY.md: actuator tables, joint ranges (degrees -> radians), block/pan geometry."""
    m = models["cfg3"]
    assert np.allclose(m.act_gear, [3, 3, 5, 1, 1, 1, 1]) and np.allclose(m.act_kp, [300, 300, 10, 1000, 1000, 1000, 1000])
    assert np.allclose(m.act_ctrlrange[2], [2.3, 4.1]) and np.allclose(m.act_forcerange[3], [-15, 35])
    assert np.allclose(m.dof_range[4], np.deg2rad([-90, 90])) and np.allclose(m.dof_range[5], [0, np.deg2rad(20)])
    assert not m.dof_limited[3]                                          # arm_flex_joint limited="false" (hsr.mjcf:131)
    assert np.allclose(m.dof_damping[:7], [2200, 2200, 100, 25, 15, 15, 15])
    blk = m.names["geom"].index("block0")
    assert np.allclose(m.geom_size[blk], [.05, .025, .017]) and m.geom_condim[blk] == 6
    assert np.allclose(m.geom_size[1], [.17, .2667, .005]) and np.allclose(m.geom_pos[1], [0, 0, .4])
    assert np.allclose(m.link_mass[-1], 1.0)                             # block mass (util.py:118)
    # every pair touching the block is condim 6, all others condim 4 (max of the two geoms)
    touches = (m.pair_geom1 == blk) | (m.pair_geom2 == blk)
    assert (m.pair_condim[touches] == 6).all() and (m.pair_condim[~touches] == 4).all()
    # chain bitmasks: hand_l chain = slide_x slide_y arm_lift arm_flex wrist_roll hand_l
    assert m.link_dofmask[5] == 0b0111111 and m.link_dofmask[6] == 0b1011111 and m.link_dofmask[7] == 0b1111110000000


def test_pair_filters(models):
    """MuJoCo pair filters: no same-link pairs, no parent-child links (unless parent is world), excludes."""
    m = models["cfg3"]
    l1, l2 = m.geom_link[m.pair_geom1], m.geom_link[m.pair_geom2]
    assert (l1 != l2).all()
    par = m.link_parent
    assert not (((par[l1] == l2) & (l2 != 0)) | ((par[l2] == l1) & (l1 != 0))).any()
    names = m.names["geom"]
    pairs = {frozenset((names[a].split(":")[0], names[b].split(":")[0])) for a, b in zip(m.pair_geom1, m.pair_geom2)}
    assert frozenset(("arm_flex_link", "hand_palm_link")) not in pairs     # world.xml:134 exclude
    assert (m.geom_type[m.pair_geom1] <= m.geom_type[m.pair_geom2]).all()


@pytest.mark.skipif(not hc.DEFAULT_REF_ROOT.exists(), reason="reference data files not on this machine")
@pytest.mark.parametrize("cfg", ["cfg1", "cfg3"])
def test_recompile_matches_committed_blob(models, cfg):
    fresh = hc.compile_model(**hc.CONFIGS[cfg])
    for k, a in models[cfg].arrays.items():
        assert np.allclose(a, fresh.arrays[k], rtol=1e-12, atol=1e-14), k


def test_numpy_mass_matrix_is_spd(models):
    for m in models.values():
        M = hc.mass_matrix(m, m.qpos0)
        assert np.allclose(M, M.T) and np.linalg.eigvalsh(M).min() > 0


def test_generated_kernel_constants_match_the_blobs():
    """hsr_env_amd/csrc/cfg_consts.h (scalar model fields as compile-time constants of the persistent-kernel instances of the
    reference configurations) is generated from the committed blobs: regenerating it must reproduce the committed file."""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    r = subprocess.run([sys.executable, str(root / "tools" / "gen_cfg_consts.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, "cfg_consts.h is stale: run python tools/gen_cfg_consts.py\n" + r.stderr
