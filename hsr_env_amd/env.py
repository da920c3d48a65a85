"""Vectorised counterpart of ``hsr.env.HSREnv`` / ``hsr.mujoco_env.MujocoEnv`` (reference:
hsr/env.py:23-209, hsr/mujoco_env.py:20-151): same constructor arguments, ``step / reset / seed /
set_state``, spaces, goal-within-geofence reward with per-substep early exit - for N envs advanced in
lockstep by libhsrsim (``BatchSim``).  With ``n_envs == 1`` every return value has the reference's scalar
shapes, so the driver loop of hsr/control.py:66-76 runs against it unchanged.

Semantics recorded in SURVEY.md section 8(a) "known reference defects": a goal is
``GoalSpec(a=<body name>, b=<point or Box(3)>, distance)``: success = |xpos(a) - b| < distance with b
written to ``mocap_pos`` at reset (the working shape of hsr/__init__.py:12); ``starts`` maps a joint name
to a Box over its qpos slice, sampled per env at reset (hsr/env.py:149-156).
"""
from __future__ import annotations

from collections import namedtuple
from pathlib import Path
from typing import Dict, List, Optional

import numpy as np

from .compiler import Model, load_config
from .spaces import Box, Space

GoalSpec = namedtuple("GoalSpec", "a b distance")     # hsr/env.py:20


def distance_between(pos1, pos2):                        # hsr/env.py:231-232
    return np.sqrt(np.sum(np.square(pos1 - pos2), axis=-1))


def block_space_to_qpos(sample4: np.ndarray) -> np.ndarray:
    """(x, y, z, yaw) of a block -> free-joint qpos (x y z qw qx qy qz); the build's reading of
    ``--block-space`` (a Box(4), hsr/util.py:33), see SURVEY.md section 8(a) defects."""
    s = np.asarray(sample4, dtype=np.float64)
    out = np.zeros(s.shape[:-1] + (7,))
    out[..., :3] = s[..., :3]
    out[..., 3] = np.cos(s[..., 3] / 2)
    out[..., 6] = np.sin(s[..., 3] / 2)
    return out


class VecHSREnv:
    metadata = {"render.modes": "rgb_array"}

    def __init__(self, xml_file=None, goals: Optional[List[GoalSpec]] = None, starts: Optional[Dict[str, Box]] = None,
                 steps_per_action: int = 300, obs_type: str = None, render: bool = False, record: bool = False,
                 record_freq: int = None, render_freq: int = None, record_path: Path = None,
                 n_envs: int = 1, model: Optional[Model] = None, sim=None, device: int = 0,
                 env_offset: int = 0, n_global: Optional[int] = None, block_space: Optional[Box] = None):
        if model is None:
            model = load_config(str(xml_file)) if xml_file is not None else None
        if model is None:
            raise IOError("File %s does not exist" % xml_file)          # hsr/mujoco_env.py:30-31
        if any([render, record, record_path, record_freq and False]):
            raise NotImplementedError("rendering / recording are outside the batched hot path (camera-free obs)")
        if obs_type not in (None, "openai"):
            raise ValueError(f"unknown obs_type {obs_type!r}")
        if obs_type == "openai" and not ({"hand_l_proximal_joint", "hand_r_proximal_joint"} <= set(model.names["joint"]) and model.block_body()):
            raise ValueError("obs_type='openai' needs the two finger joints among the DOFs and a block (hsr/env.py:58-59,90-97)")
        self.model = model
        self.n_envs = int(n_envs)
        self.env_offset, self.n_global = int(env_offset), int(n_global or n_envs)
        self.starts = dict(starts or {})
        self.block_space = block_space
        self.goals_specs = goals
        self.goals = None
        self._time_steps = np.zeros(self.n_envs, dtype=np.int64)
        self._obs_type = obs_type
        self.reward_range = -np.inf, np.inf
        self.spec = None
        self.steps_per_action = steps_per_action
        self.record_freq = record_freq or 20
        self.render_freq = render_freq or 20
        self.frame_skip = self.record_freq                              # hsr/env.py:68 passes record_freq
        # hsr/env.py:58 hard-codes 'block0' (the util.py:109 injection); cupboard-world.xml:113 names its body 'block'
        self._block_name = model.block_body() or "block0"
        self._finger_names = ["hand_l_distal_link", "hand_r_distal_link"]
        if sim is None:
            from .sim import BatchSim
            sim = BatchSim(model, self.n_envs, device=device)
        self.sim = sim
        bounds = model.act_ctrlrange.copy()
        self.action_space = Box(low=bounds[:, 0], high=bounds[:, 1], dtype=np.float32)
        self.init_qpos = model.qpos0.copy()
        self.init_qvel = np.zeros(model.nv)
        self.obs_dim = 25 if obs_type == "openai" else model.nq + model.nv
        high = np.inf * np.ones(self.obs_dim)
        self.observation_space = Box(-high, high, dtype=np.float32)
        self._goal_body, self._geofence = -1, 0.0
        self._goal_points = np.zeros((self.n_envs, 3), dtype=np.float32)
        self._parse_goals()
        self.seed()
        t, q, v = self.sim.get_state()
        self.initial_state = (t.copy(), q.copy(), v.copy())
        self._last_obs = np.concatenate([q, v], axis=1)

    # ------------------------------------------------------------------ helpers
    def _squeeze(self, x):
        return x[0] if self.n_envs == 1 else x

    def _parse_goals(self):
        """hsr/env.py:124-126,137-147: `done = all(in_range(a, b, d) for a, b, d in goals)`, an operand being a body name
        (its xpos), a point (ndarray, or a Space sampled at reset) or a callable.  On the device a goal is a pair of bodies, or a
        body and THE point: like the reference, whose reset writes the concatenation of all point operands into the single
        mocap_pos[1, 3] (hsr/env.py:169-172), at most one point operand can exist across the goals.  The goal that holds the
        point is the main term of hsr_batch_step (goal_body, geofence); body-body goals become the extra terms of
        hsr_batch_set_goals (the mocap body stands for the point there).  Callables cannot run inside the substep loop."""
        self._goal_body, self._geofence, self._point_goal, self._extra_terms = -1, 0.0, None, []
        if not self.goals_specs:
            return
        mocap_body = next((i for i, mc in enumerate(self.model.arrays["body_mocap"]) if mc), None)
        for gi, (a, b, d) in enumerate(self.goals_specs):
            for x in (a, b):
                if callable(x) and not isinstance(x, Space):
                    raise RuntimeError(f"{x} must be np.ndarray or string: callables cannot be evaluated inside the device substep loop")
                if not isinstance(x, (str, np.ndarray, Space, list, tuple)):
                    raise RuntimeError(f"{x} must be function, np.ndarray, or string")      # hsr/env.py:145
            points = [x for x in (a, b) if not isinstance(x, str)]
            if len(points) == 2:
                raise RuntimeError("a goal between two points does not depend on the simulation")
            if points:
                if self._point_goal is not None:
                    raise ValueError("the goals hold more than one point operand: mocap_pos has room for one (hsr/env.py:169-172)")
                body = a if isinstance(a, str) else b
                self._point_goal, self._goal_body, self._geofence = gi, self.model.body_id(body), float(d)
            else:
                self._extra_terms.append((self.model.body_id(a), self.model.body_id(b), float(d)))
        if len(self._extra_terms) > 4:
            raise NotImplementedError("at most four body-body goals besides the point goal")
        if self._extra_terms and not hasattr(self.sim, "set_goals"):
            raise NotImplementedError("this simulator handle does not evaluate body-body goals")
        del mocap_body

    @property
    def dt(self):
        return self.model.timestep * self.frame_skip                   # hsr/mujoco_env.py:96-98

    def seed(self, seed=None):
        self._seed = 0 if seed is None else int(seed)
        self._reset_count = 0
        self.np_random = np.random.Generator(np.random.Philox(key=self._seed))
        return [seed]

    def _global_rng(self):
        # one stream per (seed, reset index); every rank draws the global batch and keeps its shard, so a
        # sharded run reproduces the single-GPU run env for env
        return np.random.Generator(np.random.Philox(key=[self._seed, self._reset_count]))

    def _shard(self, x):
        return x[self.env_offset:self.env_offset + self.n_envs]

    # ------------------------------------------------------------------ gym surface
    def new_state(self, rng=None):
        """hsr/env.py:149-156: qpos with every joint in ``starts`` resampled (per env)."""
        rng = rng or self._global_rng()
        qpos = np.tile(self.model.qpos0, (self.n_global, 1))
        for joint, space in self.starts.items():
            assert isinstance(space, Space)
            adr = self.model.joint_qpos_addr(joint)
            start, end = adr if isinstance(adr, tuple) else (adr, adr + 1)
            qpos[:, start:end] = space.sample(self.n_global, rng=rng)
        if self.block_space is not None:
            nb = (self.model.nq - self.model.nu) // 7
            for b in range(nb):
                a = self.model.nu + 7 * b
                qpos[:, a:a + 7] = block_space_to_qpos(self.block_space.sample(self.n_global, rng=rng))
        return self._shard(qpos)

    def reset(self, mask=None):
        """sim.reset() + reset_model() (hsr/mujoco_env.py:83-85, hsr/env.py:158-177); ``mask`` selects envs."""
        rng = self._global_rng()
        self._reset_count += 1
        m = np.ones(self.n_envs, dtype=bool) if mask is None else np.asarray(mask, dtype=bool).reshape(self.n_envs)
        self._time_steps[m] = 0
        if self.goals_specs:
            if self._extra_terms and self.goals is None:
                self.sim.set_goals(self._extra_terms)               # from the first reset on (before it `goals is None`: hsr/env.py:39,125)
            self.goals = list(self.goals_specs)
            if self._point_goal is not None:
                a, b, d = self.goals_specs[self._point_goal]
                pt = b if isinstance(a, str) else a
                pts = pt.sample(self.n_global, rng=rng) if isinstance(pt, Space) else np.tile(np.asarray(pt, dtype=np.float32), (self.n_global, 1))
                pts = self._shard(np.asarray(pts, dtype=np.float32).reshape(self.n_global, 3))
                self._goal_points[m] = pts[m]
                cur = self._squeeze(self._goal_points)
                self.goals[self._point_goal] = GoalSpec(a, cur, d) if isinstance(a, str) else GoalSpec(cur, b, d)
        qpos = self.new_state(rng).astype(np.float32)
        self.sim.reset(mask=m.astype(np.uint8), qpos0=qpos, mocap=self._goal_points)
        return self._get_observation()

    def set_state(self, qpos, qvel):
        qpos = np.asarray(qpos, dtype=np.float32).reshape(self.n_envs, -1)
        qvel = np.asarray(qvel, dtype=np.float32).reshape(self.n_envs, -1)
        assert qpos.shape[1:] == (self.model.nq,) and qvel.shape[1:] == (self.model.nv,)   # hsr/mujoco_env.py:88-89
        t = self.sim.get_state()[0]
        self.sim.set_state(t, qpos, qvel)

    def state_vector(self):
        return self._get_observation()

    def _get_observation(self):
        if self._obs_type == "openai":                                  # hsr/env.py:72-110, fused on the device
            self._last_obs = self.sim.obs_openai()
            return self._squeeze(self._last_obs)
        t, q, v = self.sim.get_state()
        self._last_obs = np.concatenate([q, v], axis=1)                 # hsr/env.py:111-113
        return self._squeeze(self._last_obs)

    def step(self, action, steps=None):
        """hsr/env.py:115-135 for every env: returns (obs, reward, done, info)."""
        action = np.asarray(action, dtype=np.float32).reshape(self.n_envs, self.model.nu)
        steps = steps or self.steps_per_action
        goal_body = self._goal_body if self.goals else -1

        obs, rew, done, ns = self.sim.step(action, steps, goal_body, self._geofence)
        bad_state = getattr(self.sim, "bad_state", None)
        if bad_state is not None:
            bad, any_bad = bad_state()
            if any_bad:                                                 # mujoco_py.MujocoException on MuJoCo's divergence warnings
                from .sim import MujocoException
                raise MujocoException(f"simulation diverged in env(s) {np.flatnonzero(bad)[:8].tolist()} (non-finite or |q| > 1e10)")
        self._time_steps += 1
        if self._obs_type == "openai":
            obs = self.sim.obs_openai()
        self._last_obs = obs
        success = done
        info = {"log count": {"success": self._squeeze(success & (self._time_steps > 0))}, "substeps": self._squeeze(ns)}
        if self.n_envs == 1:
            return obs[0], float(rew[0]), bool(done[0]), info
        return obs, rew, done, info

    def in_range(self, a, b, distance):
        def parse(x):
            if callable(x):
                return x()
            if isinstance(x, np.ndarray):
                return x
            if isinstance(x, str):
                return self._squeeze(self.sim.body_xpos(self.model.body_id(x)))
            raise RuntimeError(f"{x} must be function, np.ndarray, or string")
        return distance_between(parse(a), parse(b)) < distance

    def block_pos(self):
        return self._squeeze(self.sim.body_xpos(self.model.body_id(self._block_name)))

    def gripper_pos(self):
        f1, f2 = [self.sim.body_xpos(self.model.body_id(n)) for n in self._finger_names]
        return self._squeeze((f1 + f2) / 2.)

    def get_body_com(self, body_name):
        return self._squeeze(self.sim.body_xpos(self.model.body_id(body_name)))

    def render(self, *a, **k):
        raise NotImplementedError("camera-free observations only (SURVEY.md section 2, #2)")

    def close(self):
        if getattr(self.sim, "close", None):
            self.sim.close()

    def __enter__(self):
        return self

    def __exit__(self, *args):
        self.close()


HSREnv = VecHSREnv
