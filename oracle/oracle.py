"""ctypes binding of oracle/libhsr_oracle.so (CPU fp64 restatement of mj_step).

TEST INFRASTRUCTURE ONLY - imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by the product package.  Parity with mujoco-py is unpinned (see the
header of hsr_oracle.c).
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).parent
_LIBS = {}
_REAL = {"f64": ("libhsr_oracle.so", C.c_double, np.float64), "f32": ("libhsr_oracle_f32.so", C.c_float, np.float32)}


def build(force: bool = False, real: str = "f64") -> Path:
    so = _HERE / _REAL[real][0]
    src = _HERE / "hsr_oracle.c"
    if force or not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
        subprocess.check_call(["make", "-C", str(_HERE), so.name], stdout=subprocess.DEVNULL)
    return so


def lib(real: str = "f64"):
    """real = "f32": the same restatement compiled with -DHO_REAL=float (what single precision alone does to a result; tests only)."""
    if real not in _LIBS:
        L = C.CDLL(str(build(real=real)))
        assert L.ho_real_bytes() == (8 if real == "f64" else 4)
        vp, dp, ip = C.c_void_p, C.POINTER(_REAL[real][1]), C.POINTER(C.c_int)
        L.ho_model_load.restype = vp; L.ho_model_load.argtypes = [C.c_char_p, C.c_size_t]
        L.ho_model_free.argtypes = [vp]
        L.ho_model_size.restype = C.c_int; L.ho_model_size.argtypes = [vp, C.c_int]
        L.ho_data_new.restype = vp; L.ho_data_new.argtypes = [vp]
        L.ho_data_free.argtypes = [vp]
        for f in ("ho_reset", "ho_forward", "ho_step"):
            getattr(L, f).argtypes = [vp, vp]; getattr(L, f).restype = None
        L.ho_body_xpos.argtypes = [vp, vp, C.c_int, dp]
        L.ho_env_step.restype = C.c_int
        L.ho_env_step.argtypes = [vp, vp, dp, C.c_int, C.c_int, dp, _REAL[real][1], ip]
        for f in ("qpos", "qvel", "ctrl", "qacc", "qacc_warmstart", "qacc_smooth", "qfrc_smooth",
                  "qfrc_bias", "qfrc_constraint", "mocap_pos", "xpos", "xquat", "xmat", "M", "efc_J",
                  "efc_force", "efc_aref", "efc_R", "efc_pos", "time"):
            getattr(L, "ho_" + f).restype = dp; getattr(L, "ho_" + f).argtypes = [vp]
        for f in ("ncon", "nefc", "bad", "solver_niter"):
            getattr(L, "ho_" + f).restype = C.c_int; getattr(L, "ho_" + f).argtypes = [vp]
        L.ho_contact_get.argtypes = [vp, C.c_int, dp]
        L.ho_contact_get2.argtypes = [vp, C.c_int, dp]
        L.ho_solver_stat.restype = C.c_int; L.ho_solver_stat.argtypes = [vp, C.c_int]
        L.ho_solver_trace.restype = dp; L.ho_solver_trace.argtypes = [vp]
        L.ho_set_euler_rhs.argtypes = [vp, C.c_int]; L.ho_set_euler_rhs.restype = None
        L.ho_batch_env_step.restype = C.c_int
        L.ho_batch_env_step.argtypes = [vp, C.c_int, dp, dp, dp, dp, dp, C.c_int, C.c_int, _REAL[real][1], ip, C.c_int]
        _LIBS[real] = L
    return _LIBS[real]


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double if a.dtype == np.float64 else C.c_float))


class OracleSim:
    """One fp64 environment; mirrors the slice of mujoco_py.MjSim the reference touches
    (hsr/mujoco_env.py:33-34,84,90-94,101-103; hsr/env.py:116,123,169,175-176)."""

    def __init__(self, model, real: str = "f64"):
        self.model = model
        self._dtype = _REAL[real][2]
        self._L = lib(real)
        raw = model.to_bytes()
        self._m = self._L.ho_model_load(raw, len(raw))
        assert self._m, "oracle failed to load model blob"
        self._d = self._L.ho_data_new(self._m)
        self._L.ho_reset(self._m, self._d)
        nq, nv, nu, nl = model.nq, model.nv, model.nu, model.nlink
        self.qpos = self._view("qpos", (nq,)); self.qvel = self._view("qvel", (nv,))
        self.ctrl = self._view("ctrl", (nu,)); self.qacc = self._view("qacc", (nv,))
        self.qacc_warmstart = self._view("qacc_warmstart", (nv,))
        self.qacc_smooth = self._view("qacc_smooth", (nv,)); self.qfrc_smooth = self._view("qfrc_smooth", (nv,))
        self.qfrc_bias = self._view("qfrc_bias", (nv,)); self.qfrc_constraint = self._view("qfrc_constraint", (nv,))
        self.mocap_pos = self._view("mocap_pos", (3,))
        self.xpos = self._view("xpos", (nl, 3)); self.xquat = self._view("xquat", (nl, 4))
        self.xmat = self._view("xmat", (nl, 3, 3)); self.M = self._view("M", (nv, nv))
        self._time = self._view("time", (1,))

    def _view(self, name, shape):
        ptr = getattr(self._L, "ho_" + name)(self._d)
        return np.ctypeslib.as_array(ptr, shape=(int(np.prod(shape)),)).reshape(shape)

    def __del__(self):
        try:
            self._L.ho_data_free(self._d); self._L.ho_model_free(self._m)
        except Exception:
            pass

    @property
    def time(self): return float(self._time[0])
    @time.setter
    def time(self, v): self._time[0] = v

    def set_euler_rhs(self, m_qacc: bool):
        """mj_Euler's damped solve with M qacc as right-hand side (what the HIP path integrates) instead of MuJoCo's force form."""
        self._L.ho_set_euler_rhs(self._d, int(m_qacc))

    def reset(self): self._L.ho_reset(self._m, self._d)
    def forward(self): self._L.ho_forward(self._m, self._d)
    def step(self): self._L.ho_step(self._m, self._d)

    @property
    def ncon(self): return self._L.ho_ncon(self._d)
    @property
    def nefc(self): return self._L.ho_nefc(self._d)
    @property
    def bad(self): return self._L.ho_bad(self._d)
    @property
    def solver_niter(self): return self._L.ho_solver_niter(self._d)

    def contacts(self):
        out = np.zeros((self.ncon, 17), self._dtype)
        for i in range(self.ncon):
            self._L.ho_contact_get(self._d, i, _dp(out[i]))
        return out

    def contact_cones(self):
        """Per contact: (first constraint row, dim, mu, friction[5]) - what a cost function needs besides efc()."""
        out = []
        a, b = np.zeros(17, self._dtype), np.zeros(6, self._dtype)
        for i in range(self.ncon):
            self._L.ho_contact_get(self._d, i, _dp(a)); self._L.ho_contact_get2(self._d, i, _dp(b))
            out.append((int(b[5]), int(a[15]), float(a[16]), b[:5].copy()))
        return out

    def solver_stats(self):
        """(accepted costs: after the warm-start choice, then after every Newton iteration; largest line-search evaluation count of an
        iteration; a cost-raising step was refused; number of active limit rows)."""
        n = self._L.ho_solver_stat(self._d, 0)
        tr = np.ctypeslib.as_array(self._L.ho_solver_trace(self._d), shape=(104,))[:n].copy()
        return tr, self._L.ho_solver_stat(self._d, 1), bool(self._L.ho_solver_stat(self._d, 2)), self._L.ho_solver_stat(self._d, 3)

    def efc(self):
        ne, nv = self.nefc, self.model.nv
        g = lambda n, s: self._view(n, s).copy()
        return dict(J=self._view("efc_J", (512 * nv,))[:ne * nv].reshape(ne, nv).copy(),
                    force=g("efc_force", (512,))[:ne], aref=g("efc_aref", (512,))[:ne],
                    R=g("efc_R", (512,))[:ne], pos=g("efc_pos", (512,))[:ne])

    def body_xpos(self, body_id: int):
        out = np.zeros(3, self._dtype)
        self._L.ho_body_xpos(self._m, self._d, int(body_id), _dp(out))
        return out

    def env_step(self, ctrl, nsub, goal_body=-1, goal=None, geofence=0.0):
        ctrl = np.ascontiguousarray(ctrl, dtype=self._dtype)
        goal = np.zeros(3, self._dtype) if goal is None else np.ascontiguousarray(goal, dtype=self._dtype)
        done = C.c_int(0)
        n = self._L.ho_env_step(self._m, self._d, _dp(ctrl), int(nsub), int(goal_body), _dp(goal),
                                float(geofence), C.byref(done))
        return n, bool(done.value)


def batch_env_step(model, qpos, qvel, warm, ctrl, mocap, nsub, goal_body=-1, geofence=0.0, nthreads=1):
    """n independent fp64 envs advanced one env-step (OpenMP over envs); arrays updated in place."""
    L = lib()
    raw = model.to_bytes()
    m = L.ho_model_load(raw, len(raw))
    n = qpos.shape[0]
    done = np.zeros(n, dtype=np.int32)
    for a in (qpos, qvel, warm, ctrl, mocap):
        assert a.dtype == np.float64 and a.flags.c_contiguous
    total = L.ho_batch_env_step(m, n, _dp(qpos), _dp(qvel), _dp(warm), _dp(ctrl), _dp(mocap), int(nsub),
                                int(goal_body), float(geofence), done.ctypes.data_as(C.POINTER(C.c_int)), int(nthreads))
    L.ho_model_free(m)
    return total, done
