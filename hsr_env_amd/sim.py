"""ctypes binding of libhsrsim.so (include/hsrsim.h) - the batched, GPU-resident stand-in for the
slice of ``mujoco_py`` the reference touches (hsr/mujoco_env.py:33-34,84,87-94,101-103;
hsr/env.py:116,123,144,153,169,175-176,180,184,209).

There is no CPU fallback: if the shared library is missing or no GPU is visible, construction
raises (``DependencyNotInstalled`` mirrors hsr/mujoco_env.py:12-15).
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

from .compiler import Model

_HERE = Path(__file__).parent
import os
LIB_PATH = Path(os.environ.get("HSR_LIB", _HERE / "libhsrsim.so"))


class DependencyNotInstalled(ImportError):
    """Raised when libhsrsim.so (or a GPU) is unavailable - reference: gym.error.DependencyNotInstalled
    raised by hsr/mujoco_env.py:12-15 when mujoco_py is missing."""


class MujocoException(RuntimeError):
    """Some env reached a non-finite state (reference: mujoco_py.MujocoException on MuJoCo warnings)."""


EXPORTS = [
    "hsr_last_error", "hsr_model_load", "hsr_model_destroy", "hsr_model_size", "hsr_model_timestep",
    "hsr_model_ctrlrange", "hsr_model_qpos0", "hsr_model_body_id", "hsr_model_joint_qpos_addr",
    "hsr_batch_create", "hsr_batch_destroy", "hsr_batch_size", "hsr_batch_stream", "hsr_batch_sync",
    "hsr_batch_reset", "hsr_batch_reset_dev", "hsr_batch_get_state", "hsr_batch_set_state", "hsr_batch_set_mocap",
    "hsr_batch_set_warmstart", "hsr_batch_get_warmstart", "hsr_batch_forward", "hsr_batch_step",
    "hsr_batch_step_dev", "hsr_batch_body_xpos", "hsr_batch_bad_state", "hsr_batch_get_field",
    "hsr_batch_set_profiling", "hsr_batch_last_timing", "hsr_batch_set_graph", "hsr_batch_set_persistent", "hsr_batch_is_persistent",
    "hsr_batch_obs_openai", "hsr_batch_obs_openai_dev", "hsr_batch_set_debug", "hsr_batch_cap_counts", "hsr_batch_cap_histogram", "hsr_batch_newton_trips", "hsr_batch_packing", "hsr_batch_set_schedule", "hsr_batch_set_solo", "hsr_batch_solo_handovers", "hsr_batch_set_goals",
    "hsr_batch_phase_cycles", "hsr_batch_block_times", "hsr_batch_kernel_times", "hsr_batch_set_queue", "hsr_batch_set_mpr_warm",
]

F_XPOS, F_XMAT, F_M, F_QACC, F_QACC_SMOOTH, F_QFRC_SMOOTH, F_QFRC_CONSTRAINT, F_NCON, F_NEFC, F_CONTACT, F_NITER = range(11)

_lib = None


def load_library():
    """dlopen libhsrsim.so and declare prototypes; raises DependencyNotInstalled if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise DependencyNotInstalled(
            f"{LIB_PATH} not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback for the product path.")
    L = C.CDLL(str(LIB_PATH))
    vp, fp, u8p, i32p = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_int32)
    L.hsr_last_error.restype = C.c_char_p
    L.hsr_model_load.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(vp)]
    L.hsr_model_destroy.argtypes = [vp]; L.hsr_model_destroy.restype = None
    L.hsr_model_size.argtypes = [vp, C.c_int]
    L.hsr_model_timestep.argtypes = [vp]; L.hsr_model_timestep.restype = C.c_double
    L.hsr_model_ctrlrange.argtypes = [vp, fp]
    L.hsr_model_qpos0.argtypes = [vp, fp]
    L.hsr_model_body_id.argtypes = [vp, C.c_char_p]
    L.hsr_model_joint_qpos_addr.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.hsr_batch_create.argtypes = [vp, C.c_int, C.c_int, C.POINTER(vp)]
    L.hsr_batch_destroy.argtypes = [vp]; L.hsr_batch_destroy.restype = None
    L.hsr_batch_size.argtypes = [vp]
    L.hsr_batch_stream.argtypes = [vp]; L.hsr_batch_stream.restype = vp
    L.hsr_batch_sync.argtypes = [vp]
    L.hsr_batch_reset.argtypes = [vp, u8p, fp, fp]
    L.hsr_batch_reset_dev.argtypes = [vp, vp, vp, vp]
    L.hsr_batch_get_state.argtypes = [vp, fp, fp, fp]
    L.hsr_batch_set_state.argtypes = [vp, fp, fp, fp]
    L.hsr_batch_set_mocap.argtypes = [vp, fp]
    L.hsr_batch_set_warmstart.argtypes = [vp, fp]
    L.hsr_batch_get_warmstart.argtypes = [vp, fp]
    L.hsr_batch_forward.argtypes = [vp]
    L.hsr_batch_step.argtypes = [vp, fp, C.c_int, C.c_int, C.c_float, fp, fp, u8p, i32p]
    L.hsr_batch_step_dev.argtypes = [vp, vp, C.c_int, C.c_int, C.c_float, vp, vp, vp, vp]
    L.hsr_batch_body_xpos.argtypes = [vp, C.c_int, fp]
    L.hsr_batch_obs_openai.argtypes = [vp, C.POINTER(C.c_int), fp]
    L.hsr_batch_obs_openai_dev.argtypes = [vp, C.POINTER(C.c_int), C.c_void_p]
    L.hsr_batch_bad_state.argtypes = [vp, u8p]
    L.hsr_batch_get_field.argtypes = [vp, C.c_int, fp]
    L.hsr_batch_set_profiling.argtypes = [vp, C.c_int]
    L.hsr_batch_last_timing.argtypes = [vp, fp, fp, C.POINTER(C.c_int)]
    L.hsr_batch_kernel_times.argtypes = [vp, fp, C.c_int]
    L.hsr_batch_set_queue.argtypes = [vp, C.c_int, C.c_int]
    L.hsr_batch_set_mpr_warm.argtypes = [vp, C.c_int]
    L.hsr_batch_set_graph.argtypes = [vp, C.c_int]
    L.hsr_batch_set_persistent.argtypes = [vp, C.c_int]
    L.hsr_batch_is_persistent.argtypes = [vp]
    L.hsr_batch_set_debug.argtypes = [vp, C.c_int]
    L.hsr_batch_set_schedule.argtypes = [vp, C.c_int]
    L.hsr_batch_set_goals.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), fp]
    L.hsr_batch_cap_counts.argtypes = [vp, C.POINTER(C.c_ulonglong)]
    L.hsr_batch_cap_histogram.argtypes = [vp, C.POINTER(C.c_ulonglong)]
    L.hsr_batch_newton_trips.argtypes = [vp, C.POINTER(C.c_int32)]
    L.hsr_batch_packing.argtypes = [vp, C.POINTER(C.c_int32)]
    L.hsr_batch_set_solo.argtypes = [vp, C.c_int, C.c_float]
    L.hsr_batch_solo_handovers.argtypes = [vp, C.POINTER(C.c_int)]
    _lib = L
    return L


def _fp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))


def _f32(a, shape=None):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None:
        assert a.shape == tuple(shape), f"expected shape {tuple(shape)}, got {a.shape}"
    return a


def _check(L, rc):
    if rc == 0:
        return
    msg = L.hsr_last_error().decode()
    if rc == -3:
        raise DependencyNotInstalled(f"libhsrsim: {msg}")
    if rc == -2:
        raise IOError(f"libhsrsim: {msg}")
    if rc == -5:
        raise MujocoException(f"libhsrsim: {msg}")
    if rc == -4:
        raise KeyError(msg)
    raise AssertionError(f"libhsrsim: {msg}")


class BatchSim:
    """N independent simulations advanced in lockstep on one GPU.

    Mirrors, per env, ``mujoco_py.MjSim``: ``step/forward/reset/get_state/set_state`` and the
    ``sim.data`` fields the reference reads or writes (ctrl, qpos, qvel, mocap_pos, body xpos)."""

    def __init__(self, model: Model, n_envs: int, device: int = 0):
        self.model = model
        self.n = int(n_envs)
        self._L = L = load_library()
        raw = model.to_bytes()
        self._m = C.c_void_p()
        _check(L, L.hsr_model_load(raw, len(raw), C.byref(self._m)))
        self._b = C.c_void_p()
        _check(L, L.hsr_batch_create(self._m, self.n, int(device), C.byref(self._b)))
        self.nq, self.nv, self.nu = model.nq, model.nv, model.nu
        self.device = device

    def close(self):
        if getattr(self, "_b", None):
            self._L.hsr_batch_destroy(self._b); self._b = None
        if getattr(self, "_m", None):
            self._L.hsr_model_destroy(self._m); self._m = None

    __del__ = close

    # -- sim.reset() (+ the qpos/mocap writes of reset_model) then forward
    def reset(self, mask=None, qpos0=None, mocap=None):
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        q = _f32(qpos0, (self.n, self.nq)); mc = _f32(mocap, (self.n, 3))
        _check(self._L, self._L.hsr_batch_reset(self._b, None if m is None else m.ctypes.data_as(C.POINTER(C.c_uint8)), _fp(q), _fp(mc)))

    def reset_dev(self, d_mask, d_qpos0, d_mocap):
        """Device-pointer masked reset (d_mask None/0 -> envs whose done flag was latched); async."""
        _check(self._L, self._L.hsr_batch_reset_dev(self._b, d_mask, d_qpos0, d_mocap))

    def forward(self):
        _check(self._L, self._L.hsr_batch_forward(self._b))

    def get_state(self):
        t = np.empty(self.n, np.float32); q = np.empty((self.n, self.nq), np.float32); v = np.empty((self.n, self.nv), np.float32)
        _check(self._L, self._L.hsr_batch_get_state(self._b, _fp(t), _fp(q), _fp(v)))
        return t, q, v

    def set_state(self, time=None, qpos=None, qvel=None):
        t = _f32(time, (self.n,)); q = _f32(qpos, (self.n, self.nq)); v = _f32(qvel, (self.n, self.nv))
        _check(self._L, self._L.hsr_batch_set_state(self._b, _fp(t), _fp(q), _fp(v)))

    def set_mocap(self, mocap):
        _check(self._L, self._L.hsr_batch_set_mocap(self._b, _fp(_f32(mocap, (self.n, 3)))))

    def set_warmstart(self, w):
        _check(self._L, self._L.hsr_batch_set_warmstart(self._b, _fp(_f32(w, (self.n, self.nv)))))

    def get_warmstart(self):
        w = np.empty((self.n, self.nv), np.float32)
        _check(self._L, self._L.hsr_batch_get_warmstart(self._b, _fp(w)))
        return w

    def step(self, ctrl, n_substeps, goal_body=-1, geofence=0.0):
        """HSREnv.step for all envs -> (obs[N,nq+nv], reward[N], done[N] bool, nsteps[N])."""
        c = _f32(ctrl, (self.n, self.nu))
        obs = np.empty((self.n, self.nq + self.nv), np.float32); rew = np.empty(self.n, np.float32)
        done = np.empty(self.n, np.uint8); ns = np.empty(self.n, np.int32)
        _check(self._L, self._L.hsr_batch_step(self._b, _fp(c), int(n_substeps), int(goal_body), float(geofence), _fp(obs), _fp(rew),
                                               done.ctypes.data_as(C.POINTER(C.c_uint8)), ns.ctypes.data_as(C.POINTER(C.c_int32))))
        return obs, rew, done.astype(bool), ns

    def step_dev(self, d_ctrl, n_substeps, goal_body, geofence, d_obs=None, d_reward=None, d_done=None, d_nsteps=None):
        """Device-pointer variant (ints from tensor.data_ptr()); asynchronous on the batch stream."""
        _check(self._L, self._L.hsr_batch_step_dev(self._b, d_ctrl, int(n_substeps), int(goal_body), float(geofence),
                                                   d_obs, d_reward, d_done, d_nsteps))

    def sync(self):
        _check(self._L, self._L.hsr_batch_sync(self._b))

    def stream_ptr(self) -> int:
        """hipStream_t of the batch (wrap with torch.cuda.ExternalStream to order torch ops after steps)."""
        return int(self._L.hsr_batch_stream(self._b))

    def body_xpos(self, body_id: int):
        out = np.empty((self.n, 3), np.float32)
        _check(self._L, self._L.hsr_batch_body_xpos(self._b, int(body_id), _fp(out)))
        return out

    def openai_ids(self, finger_bodies=("hand_l_distal_link", "hand_r_distal_link"), object_body=None,
                   finger_joints=("hand_l_proximal_joint", "hand_r_proximal_joint")):
        """Body ids / joint addresses the fused 'openai' observation needs (names of hsr/env.py:58-59,90-97)."""
        m = self.model
        jn, qa, da = m.names["joint"], m.meta["joint_qposadr"], m.meta["joint_dofadr"]
        ids = [m.body_id(finger_bodies[0]), m.body_id(finger_bodies[1]), m.body_id(object_body or m.block_body())]
        ids += [qa[jn.index(j)][0] for j in finger_joints] + [da[jn.index(j)] for j in finger_joints]
        return (C.c_int * 7)(*ids)

    def obs_openai(self, ids=None):
        """[N, 25] 'openai' observation (include/hsrsim.h: hsr_batch_obs_openai)."""
        out = np.empty((self.n, 25), np.float32)
        _check(self._L, self._L.hsr_batch_obs_openai(self._b, ids or self.openai_ids(), _fp(out)))
        return out

    def obs_openai_dev(self, d_out, ids=None):
        _check(self._L, self._L.hsr_batch_obs_openai_dev(self._b, ids or self.openai_ids(), C.c_void_p(int(d_out))))

    def bad_state(self):
        out = np.empty(self.n, np.uint8)
        rc = self._L.hsr_batch_bad_state(self._b, out.ctypes.data_as(C.POINTER(C.c_uint8)))
        if rc not in (0, -5):
            _check(self._L, rc)
        return out.astype(bool), rc == -5

    def get_field(self, field: int):
        m = self.model
        shapes = {F_XPOS: (self.n, m.nlink, 3), F_XMAT: (self.n, m.nlink, 3, 3), F_M: (self.n, m.nv, m.nv),
                  F_QACC: (self.n, m.nv), F_QACC_SMOOTH: (self.n, m.nv), F_QFRC_SMOOTH: (self.n, m.nv),
                  F_QFRC_CONSTRAINT: (self.n, m.nv), F_NCON: (self.n,), F_NEFC: (self.n,), F_NITER: (self.n,),
                  F_CONTACT: (self.n, m.nslot, 7)}
        out = np.empty(shapes[field], np.float32)
        _check(self._L, self._L.hsr_batch_get_field(self._b, field, _fp(out)))
        return out

    def set_profiling(self, on):
        """True / 1: time the next step (synchronises); 2: log every launch of the persistent kernel (no synchronisation)."""
        _check(self._L, self._L.hsr_batch_set_profiling(self._b, int(on)))

    def set_mpr_warm(self, on: bool):
        """Portal warm start of the convex-pair narrowphase (include/hsrsim.h: hsr_batch_set_mpr_warm)."""
        _check(self._L, self._L.hsr_batch_set_mpr_warm(self._b, int(on)))

    def set_queue(self, mode: int, chunk: int = 0):
        """Work queue of the persistent kernel: mode -1 automatic, 0 off, 1 on; chunk = substeps per round (0 keeps the current one)."""
        _check(self._L, self._L.hsr_batch_set_queue(self._b, int(mode), int(chunk)))

    def kernel_times(self, cap: int = 4096):
        """Durations (ms) of the persistent-kernel launches logged since the last call (set_profiling(2)); synchronises."""
        buf = (C.c_float * cap)()
        n = self._L.hsr_batch_kernel_times(self._b, buf, cap)
        if n < 0:
            _check(self._L, n)
        if n > cap:
            raise ValueError(f"{n} launches were logged, the buffer holds {cap}: pass cap >= the number of launches since the last call")
        return np.array(buf[:n], dtype=np.float64)

    def set_graph(self, on: bool):
        _check(self._L, self._L.hsr_batch_set_graph(self._b, int(on)))

    def set_persistent(self, on: bool) -> bool:
        return bool(self._L.hsr_batch_set_persistent(self._b, int(on)))

    def is_persistent(self) -> bool:
        return bool(self._L.hsr_batch_is_persistent(self._b))

    def kernel_flags(self) -> int:
        """hsr_batch_is_persistent's bit mask: 1 = whole env-step in the persistent kernel, 2 = the instance carries the model's scalars as
        compile-time constants, 4 = and its kinematic tree (csrc/kin3.h)."""
        return int(self._L.hsr_batch_is_persistent(self._b))

    def set_debug(self, on):
        """Also store the contact counts / solver counters of each env's last substep during step() (persistent kernel).
        `on` may be a bit mask: 1 = store, 2 / 4 = test hooks (J v per contact / PSD-majorant Newton steps), include/hsrsim.h."""
        _check(self._L, self._L.hsr_batch_set_debug(self._b, int(on)))

    def set_goals(self, terms):
        """Further goal terms [(body_a, body_b, distance), ...] AND-ed with the main term of step() (include/hsrsim.h)."""
        n = len(terms)
        a = (C.c_int * max(n, 1))(*[int(t[0]) for t in terms]); b = (C.c_int * max(n, 1))(*[int(t[1]) for t in terms])
        d = (C.c_float * max(n, 1))(*[float(t[2]) for t in terms])
        _check(self._L, self._L.hsr_batch_set_goals(self._b, n, a, b, d))

    def set_solo(self, servers: int, trips: float = 0.0) -> bool:
        """Solo servers of the persistent kernel (include/hsrsim.h): `servers` workgroups run hard envs alone; 0 = off.  Results then
        depend on the hand-over at rounding level (another summation order of the contact terms), not bit for bit.  False when the model's kernel instance has no server path."""
        rc = self._L.hsr_batch_set_solo(self._b, int(servers), float(trips))
        if rc < 0:
            _check(self._L, rc)
        return rc == 0

    def solo_handovers(self) -> int:
        """Envs handed over to solo servers by the last step() (persistent kernel); synchronises."""
        out = C.c_int(0)
        _check(self._L, self._L.hsr_batch_solo_handovers(self._b, C.byref(out)))
        return int(out.value)

    def set_schedule(self, on: bool):
        """Wave packing of the persistent kernel by env hardness (default on); never changes a result."""
        _check(self._L, self._L.hsr_batch_set_schedule(self._b, int(on)))

    def cap_counts(self):
        """(contact-cap hits, row-cap hits, item-cap hits, env-substeps executed) since the last call."""
        out = (C.c_ulonglong * 4)()
        _check(self._L, self._L.hsr_batch_cap_counts(self._b, out))
        return tuple(int(x) for x in out)

    def cap_histogram(self):
        """Row-cap events by how many rows beyond njmax the env wanted (bins of 8 rows, last bin open); cleared by the call."""
        out = (C.c_ulonglong * 8)()
        _check(self._L, self._L.hsr_batch_cap_histogram(self._b, out))
        return [int(x) for x in out]

    def newton_trips(self):
        """Newton iterations of every env over the last (up to) 100 substeps of the previous step() (persistent kernel)."""
        out = np.empty(self.n, np.int32)
        _check(self._L, self._L.hsr_batch_newton_trips(self._b, out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out

    def packing(self, envs_per_wave: int):
        """env held by every lane group of every task of the last persistent launch (-1: empty), [tasks, envs_per_wave]."""
        if envs_per_wave not in (2, 4):
            raise ValueError("envs_per_wave is 4 (16 lanes per env) or 2 (32 lanes per env)")
        # the library writes ceil(N / epw) * epw entries with ITS envs-per-wave: the buffer covers either value whatever the caller passed
        out = np.full(self.n + 64, -1, np.int32)
        _check(self._L, self._L.hsr_batch_packing(self._b, out.ctypes.data_as(C.POINTER(C.c_int32))))
        slots = (self.n + envs_per_wave - 1) // envs_per_wave * envs_per_wave
        return out[:slots].reshape(-1, envs_per_wave)

    def last_timing(self):
        tot = C.c_float(0); k = (C.c_float * 3)(); n = (C.c_int * 3)()
        _check(self._L, self._L.hsr_batch_last_timing(self._b, C.byref(tot), k, n))
        return float(tot.value), [float(x) for x in k], [int(x) for x in n]
