"""Newton iteration counts of step-length policies on the 250 hard solver states (tests/golden/solver_states.npz), on the numpy restatement of the
cost (tests/test_oracle_optimality.py: Problem) - CPU only: is the exact line search what costs the hard envs their iterations?
  exact    exact line search on phi' (what oracle and kernels do)
  armijo   the full Newton step whenever it lowers the cost by 1e-4 of the predicted decrease, else the exact search
  halving  the full step, halved until it lowers the cost
Stop: 1/2 |phi'(0)| scale < 1e-8 (the kernels' criterion).  Start: the cheaper of qacc_smooth and the warm start."""
import sys
from pathlib import Path
import numpy as np
from scipy import optimize
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import test_oracle_optimality as too


def exact_alpha(P, a, s):
    f = lambda al: float(P.grad(a + al * s) @ s)
    if f(1.0) <= 0:
        hi = 2.0
        while f(hi) < 0 and hi < 64:
            hi *= 2
        return optimize.brentq(f, 1.0, hi, xtol=1e-10) if f(hi) > 0 else hi
    return optimize.brentq(f, 0.0, 1.0, xtol=1e-10)


def solve(P, a0, policy, maxit=50):
    a = a0.copy()
    evals = 0
    for it in range(maxit):
        g = P.grad(a)
        H = P.hess(a)
        s = np.linalg.solve(H, -g)
        dec = -float(g @ s)
        if 0.5 * dec * P.scale < 1e-8:
            return it, evals, a
        if policy == "exact":
            al = exact_alpha(P, a, s); evals += 2
        elif policy == "armijo":
            evals += 1
            if P.cost(a + s) <= P.cost(a) - 1e-4 * dec:
                al = 1.0
            else:
                al = exact_alpha(P, a, s); evals += 2
        else:
            al = 1.0; evals += 1
            c0 = P.cost(a)
            while P.cost(a + al * s) > c0 and al > 1e-6:
                al *= 0.5; evals += 1
        a = a + al * s
    return maxit, evals, a


res = {}
for i, rg, m, q, v, w, c in too.load_states():
    o = too.oracle_at(m, q, v, w, c)
    P = too.Problem(o)
    a0 = w.copy() if P.cost(w) < P.cost(P.qas) else P.qas.copy()
    ref = too.minimise(P, P.qas.copy())
    for pol in ("exact", "armijo", "halving"):
        it, ev, a = solve(P, a0, pol)
        err = np.linalg.norm(a - ref) / (1 + np.linalg.norm(ref))
        res.setdefault((rg, pol), []).append((it, ev, err, o.solver_niter))
for (rg, pol), r in sorted(res.items()):
    r = np.array(r)
    print(f"{rg:9s} {pol:8s}: iterations mean {r[:, 0].mean():5.2f} max {int(r[:, 0].max()):3d}  cost evaluations mean {r[:, 1].mean():5.2f}  worst |a - a*| {r[:, 2].max():.1e}   (oracle's own count: {r[:, 3].mean():.2f})")
