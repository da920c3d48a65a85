"""Driver loop of the reference (hsr/control.py:48-86) without the glfw viewer: zero action (or random
with --random-actions), ``if done: env.reset()``, for N envs at once.

    python -m hsr_env_amd.control --block-space "(-.1,.1)(-.2,.2)(.422,.422)(-3.14,3.14)" \\
        --steps-per-action=300 --geofence=.05 --goal-space "(-.1,.1)(-.2,.2)(.422,.422)" \\
        --use-dof slide_x --use-dof slide_y --n-blocks 1 --n-envs 4096 --env-steps 10
"""
from __future__ import annotations

import argparse
import time

import numpy as np

from . import util
from .env import VecHSREnv
from .spaces import space_to_size


class ControlHSREnv(VecHSREnv):
    def control_agent(self, random_actions=False):
        action = np.zeros((self.n_envs, space_to_size(self.action_space)), dtype=np.float32)
        if random_actions:
            action = self.action_space.sample(self.n_envs, rng=self.np_random)
        s, r, t, i = self.step(action)
        return t


def run(env, env_steps=0, random_actions=False):
    """The reference's loop (hsr/control.py:66-76: `if done: env.reset()`, `done = env.control_agent()`) over all envs of the handle at once:
    the envs that finished are reset by mask, the others go on.  env_steps <= 0 loops for ever like the reference.  Returns (env-steps, seconds)."""
    n_envs = env.n_envs
    done = np.zeros(n_envs, dtype=bool)
    k, t0 = 0, time.perf_counter()
    while env_steps <= 0 or k < env_steps:
        if np.any(done):
            env.reset(mask=np.atleast_1d(done))
        done = np.atleast_1d(env.control_agent(random_actions))
        k += 1
    dt = time.perf_counter() - t0
    print(f"{k} env-steps x {n_envs} envs in {dt:.3f} s -> {k * n_envs / dt:.1f} env-steps/s")
    return k, dt


def main(env_args, n_envs=1, env_steps=0, random_actions=False):
    env = ControlHSREnv(n_envs=n_envs, **env_args)
    env.reset()
    try:
        run(env, env_steps, random_actions)
    finally:
        env.close()


if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    wrapper_parser = parser.add_argument_group('wrapper_args')
    env_parser = parser.add_argument_group('env_args')
    util.add_env_args(env_parser)
    util.add_wrapper_args(wrapper_parser)
    parser.add_argument('--n-envs', type=int, default=1)
    parser.add_argument('--env-steps', type=int, default=0)
    parser.add_argument('--random-actions', action='store_true')
    args = util.hierarchical_parse_args(parser)
    util.env_wrapper(main)(**args)
