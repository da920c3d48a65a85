"""Test helper: a CPU stand-in with the BatchSim interface, built on the fp64 oracle.  Lives under tests/
(the product never imports it); lets the host-side logic (VecHSREnv, sharding, all-gather) run without a GPU."""
import numpy as np

from oracle.oracle import OracleSim


class OracleBatchSim:
    def __init__(self, model, n_envs, device=0):
        self.model, self.n = model, n_envs
        self.nq, self.nv, self.nu = model.nq, model.nv, model.nu
        self.sims = [OracleSim(model) for _ in range(n_envs)]
        for s in self.sims:
            s.forward()

    def reset(self, mask=None, qpos0=None, mocap=None):
        for e, s in enumerate(self.sims):
            if mask is not None and not mask[e]:
                continue
            s.reset()
            if qpos0 is not None:
                s.qpos[:] = qpos0[e]
            if mocap is not None:
                s.mocap_pos[:] = mocap[e]
            s.forward()

    def forward(self):
        for s in self.sims:
            s.forward()

    def get_state(self):
        return (np.array([s.time for s in self.sims], np.float32), np.array([s.qpos for s in self.sims], np.float32),
                np.array([s.qvel for s in self.sims], np.float32))

    def set_state(self, time=None, qpos=None, qvel=None):
        for e, s in enumerate(self.sims):
            if time is not None: s.time = float(time[e])
            if qpos is not None: s.qpos[:] = qpos[e]
            if qvel is not None: s.qvel[:] = qvel[e]
            s.forward()

    def step(self, ctrl, n_substeps, goal_body=-1, geofence=0.0):
        obs = np.zeros((self.n, self.nq + self.nv), np.float32); rew = np.zeros(self.n, np.float32)
        done = np.zeros(self.n, bool); ns = np.zeros(self.n, np.int32)
        for e, s in enumerate(self.sims):
            k, d = s.env_step(np.asarray(ctrl[e], np.float64), n_substeps, goal_body, s.mocap_pos.copy(), geofence)
            obs[e] = np.concatenate([s.qpos, s.qvel]); rew[e] = float(d); done[e] = d; ns[e] = k
        return obs, rew, done, ns

    def body_xpos(self, body_id):
        return np.array([s.body_xpos(body_id) for s in self.sims], np.float32)

    def obs_openai(self, ids=None):
        # CPU stand-in: valid after reset / set_state / forward (the forward-pass state is the current state)
        return np.array([openai_obs_reference(self.model, s.qpos.copy(), s.qvel.copy(), s.qpos, s.qvel) for s in self.sims], np.float32)

    def close(self):
        self.sims = []


def mat2euler(mat):
    """Checker-side restatement of the reference's rotation -> Euler convention (hsr/env.py:256-272): what k_obs_openai must
    reproduce for the object's rotation."""
    mat = np.asarray(mat, dtype=np.float64)
    cy = np.hypot(mat[..., 2, 2], mat[..., 1, 2])
    ok = cy > 4 * np.finfo(np.float64).eps
    z = np.where(ok, -np.arctan2(mat[..., 0, 1], mat[..., 0, 0]), -np.arctan2(-mat[..., 1, 0], mat[..., 1, 1]))
    y = -np.arctan2(-mat[..., 0, 2], cy)
    x = np.where(ok, -np.arctan2(mat[..., 1, 2], mat[..., 2, 2]), 0.0)
    return np.stack([x, y, z], axis=-1)


def openai_obs_reference(m, q_fk, v_fk, q_now, v_now, finger_bodies=("hand_l_distal_link", "hand_r_distal_link"),
                         finger_joints=("hand_l_proximal_joint", "hand_r_proximal_joint"), object_body=None):
    """numpy fp64 restatement of the 'openai' observation (hsr/env.py:72-110 with the intent of SURVEY.md 8a-5): body poses
    and velocities from the forward pass at (q_fk, v_fk) (velocity of a body origin = point Jacobian x qvel, the content
    of mj_objectVelocity), joint values from the current state, dt = timestep."""
    from hsr_env_amd import compiler as hc
    dt = m.timestep
    xpos, xquat = hc.link_kinematics(m, np.asarray(q_fk, np.float64))
    ang, lin, anchor = hc.dof_motion(m, xpos, xquat, np.asarray(q_fk, np.float64))

    def body(name):
        b = m.body_id(name)
        l = int(m.arrays["body_link"][b])
        R = hc.quat_to_mat(xquat[l])
        p = xpos[l] + R @ m.arrays["body_pos"][b]
        jp, jr = hc.point_jacobian(m, ang, lin, anchor, l, p)
        return p, jp @ v_fk, jr @ v_fk, R @ hc.quat_to_mat(m.arrays["body_quat"][b])

    pl, vl, _, _ = body(finger_bodies[0]); pr, vr, _, _ = body(finger_bodies[1])
    po, vo, wo, Ro = body(object_body or m.block_body())
    grip, gvel = (pl + pr) / 2, .5 * (vl + vr) * dt
    jn = m.names["joint"]
    qa = [m.meta["joint_qposadr"][jn.index(j)][0] for j in finger_joints]
    da = [m.meta["joint_dofadr"][jn.index(j)] for j in finger_joints]
    return np.concatenate([grip, po, po - grip, np.asarray(q_now)[qa], mat2euler(Ro), vo * dt - gvel, wo * dt, gvel,
                           dt * .5 * np.asarray(v_now)[da]])
