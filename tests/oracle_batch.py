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

    def close(self):
        self.sims = []
