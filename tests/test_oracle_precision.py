"""What single precision alone does to the contact list in the pinch regime (CPU): the oracle compiled with
-DHO_REAL=float (oracle/Makefile: libhsr_oracle_f32.so) against the fp64 oracle at the same states - the block dropped between the fingers,
two to five mesh <-> box contacts per env (hsr/models/hsr.mjcf:180,192,217,228 against the block of hsr/util.py:115-125).  libccd's MPR measures
the penetration to the final portal TRIANGLE; fp32 and fp64 do not always end on the same triangle of a face, and then depth and normal differ
by far more than rounding.  No kernel is involved here: the share of envs beyond the stage tolerances (depth 1e-5, normal 2e-3, position 2e-4)
is a property of the ALGORITHM in fp32, and it is the size of the allowance tests/test_gpu_hotpath.py grants the kernels in this regime
(test_pinched_block_contacts_follow_the_oracle; its GPU twin test_pinch_allowance_is_precision shows that the kernel and the fp32 oracle leave
the fp64 oracle in the SAME envs)."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from hsr_env_amd.compiler import load_config          # noqa: E402
from oracle.oracle import OracleSim                    # noqa: E402
from test_gpu_parity import contact_mismatch, random_states          # noqa: E402


def test_single_precision_moves_pinch_contacts_beyond_the_stage_tolerances():
    m = load_config("cfg3")
    n = 96
    rng = np.random.default_rng(3)
    q, v, ctrl = random_states(m, n, rng)
    bl, br = m.body_id("hand_l_distal_link"), m.body_id("hand_r_distal_link")
    a = m.free_joint_qadrs()[0]
    beyond, nconvex, worst = [], 0, 0.0
    fn, g1, g2 = m.arrays["pair_fn"], m.arrays["pair_geom1"], m.arrays["pair_geom2"]
    convex = {(int(g1[p]), int(g2[p])) for p in range(m.npair) if fn[p] == 3}
    for e in range(n):
        o = OracleSim(m); o.qpos[:] = q[e]; o.forward()
        q[e, a:a + 3] = 0.5 * (o.body_xpos(bl) + o.body_xpos(br)) + rng.uniform(-0.01, 0.01, 3)
        quat = rng.normal(size=4); q[e, a + 3:a + 7] = quat / np.linalg.norm(quat)
        o.qpos[:] = q[e]; o.qvel[:] = 0; o.ctrl[:] = ctrl[e]
        o.step()
        qs = o.qpos.astype(np.float32).astype(np.float64); vs = o.qvel.astype(np.float32).astype(np.float64)          # a state both precisions can hold
        lists = []
        for real in ("f64", "f32"):
            p = OracleSim(m, real)
            p.qpos[:] = qs; p.qvel[:] = vs; p.ctrl[:] = ctrl[e]
            p.forward()
            lists.append(p.contacts().astype(np.float64))
        nconvex += sum((int(r[13]), int(r[14])) in convex for r in lists[0])
        slots = np.zeros((m.nslot, 7)); slots[:, 6] = 1.0
        for pp in range(m.npair):
            rows = lists[1][(lists[1][:, 13] == g1[pp]) & (lists[1][:, 14] == g2[pp])] if len(lists[1]) else lists[1]
            for i, r in enumerate(rows[:int(m.pair_slot[pp + 1]) - int(m.pair_slot[pp])]):
                slots[int(m.pair_slot[pp]) + i] = np.r_[r[0:3], r[3:6], r[12]]
        why = contact_mismatch(m, slots, lists[0])
        if why is not None:
            beyond.append((e, why))
            if why.split()[0] == "depth":
                worst = max(worst, float(why.split()[1]))
    print(f"fp32 oracle vs fp64 oracle, pinch regime: {len(beyond)} of {n} envs beyond the stage tolerances ({nconvex} convex contacts): {beyond}")
    # measured: 12 of 96 envs (317 convex contacts), depth differences up to 6e-4, normals up to 3e-2
    assert nconvex > 200
    assert 4 <= len(beyond) <= 0.2 * n, beyond
    assert all(w.split()[0] in ("depth", "normal", "position") for _, w in beyond), beyond          # never a different contact count
    assert worst > 1e-4          # far beyond rounding: another triangle, not another bit
