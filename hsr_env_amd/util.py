"""CLI flags of the reference (hsr/util.py:16-81, rl_utils/argparse.py:10-73) for the build's own driver.

``--use-dof`` and ``--n-blocks`` select a compiled model: the committed blobs cover BASELINE's four
configurations; any other combination is compiled on the fly when the reference's data files are on disk."""
from __future__ import annotations

import argparse
from collections import namedtuple
import re
from pathlib import Path

import numpy as np

from .compiler import ALL_DOFS, CONFIGS, DEFAULT_REF_ROOT, compile_model, load_config
from .env import GoalSpec
from .spaces import Box


def make_box(*tuples):
    low, high = map(np.array, zip(*[(map(float, m)) for m in tuples]))
    return Box(low=low, high=high, dtype=np.float32)


def parse_space(dim: int):
    """rl_utils/argparse.py:62-73: '(lo,hi)(lo,hi)...' with exactly ``dim`` pairs."""
    def _parse_space(arg: str):
        regex = re.compile(r'\((-?[\.\d]+),(-?[\.\d]+)\)')
        matches = regex.findall(arg)
        if len(matches) != dim:
            raise argparse.ArgumentTypeError(f'Arg {arg} must have {dim} substrings matching pattern {regex.pattern}.')
        return make_box(*matches)
    return _parse_space


def add_env_args(parser):
    parser.add_argument('--obs-type', type=str, default=None)
    parser.add_argument('--render', action='store_true')
    parser.add_argument('--render-freq', type=int, default=None)
    parser.add_argument('--record', action='store_true')
    parser.add_argument('--record-freq', type=int, default=None)
    parser.add_argument('--record-path', type=Path, default=None)
    parser.add_argument('--steps-per-action', type=int, required=True)


def add_wrapper_args(parser):
    parser.add_argument('--block-space', type=parse_space(dim=4))
    parser.add_argument('--goal-space', type=parse_space(dim=3), required=True)
    parser.add_argument('--xml-file', type=Path, default='models/world.xml')
    parser.add_argument('--set-xml', type=xml_setter, action='append')
    parser.add_argument('--use-dof', type=str, action='append', default=[])
    parser.add_argument('--geofence', type=float, required=True)
    parser.add_argument('--n-blocks', type=int, default=0)


def hierarchical_parse_args(parser, argv=None):
    """rl_utils/argparse.py:10-47 (with the Python >= 3.10 group title 'options')."""
    args = parser.parse_args(argv)
    out = {}
    for group in parser._action_groups:
        d = {a.dest: getattr(args, a.dest, None) for a in group._group_actions if a.dest != 'help'}
        if group.title in ('positional arguments',):
            continue
        if group.title in ('optional arguments', 'options'):
            out.update(d)
        else:
            out[group.title] = d
    return out


XMLSetter = namedtuple('XMLSetter', 'path value')          # hsr/util.py:84


def xml_setter(arg: str):
    """hsr/util.py:43-44: `--set-xml path,value`."""
    return XMLSetter(*arg.split(','))


def model_for(dofs, n_blocks, xml_file='models/world.xml', set_xml=()):
    """The committed blob when the request is one of the compiled configs; otherwise the model compiler on the MJCF / STL
    data files (present where the reference tree is; --set-xml always takes this route)."""
    dofs = [d for d in ALL_DOFS if d in dofs]
    for name, kw in CONFIGS.items():
        if not set_xml and kw['dofs'] == dofs and kw['n_blocks'] == n_blocks and str(xml_file) == kw.get('xml_file', 'models/world.xml'):
            return load_config(name)
    if not DEFAULT_REF_ROOT.exists():
        raise IOError(f"no compiled model for dofs={dofs} n_blocks={n_blocks} and the MJCF/STL data files are not on disk")
    return compile_model(dofs=dofs, n_blocks=n_blocks, xml_file=str(xml_file), set_xml=list(set_xml))


def env_wrapper(func):
    """hsr/util.py:53-81: turns wrapper_args into HSREnv kwargs.  The reference builds
    GoalSpec(a=block_space, b=goal_space) which cannot run (SURVEY.md 8a defects); the build uses the
    working shape GoalSpec('block0', goal_space, geofence) and block_space as the blocks' reset pose."""
    def _wrapper(set_xml, use_dof, n_blocks, goal_space, xml_file, geofence, env_args, block_space, **kwargs):
        model = model_for(use_dof, n_blocks, xml_file, set_xml or ())
        block = model.block_body()
        goals = [GoalSpec(a=block, b=goal_space, distance=geofence)] if block else None
        env_args = dict(env_args)
        env_args.update(goals=goals, model=model, starts={}, block_space=block_space if block else None)
        return func(env_args=env_args, **kwargs)

    def new_function(wrapper_args, **kwargs):
        return _wrapper(**wrapper_args, **kwargs)
    return new_function
