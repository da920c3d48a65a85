// Soft-constraint impedance shared by the constraint assembly of the solver (solve_body.inc): mj_makeImpedance behind
// `self.sim.step()` (hsr/env.py:123), SURVEY.md section 8 a-2.4.
#pragma once
#include "devmath.h"
#include "model.h"

// GEN = false (the kernel instances of models whose solimp powers are all 1 or 2 - every committed one; cfg_consts.h solimp_general): the powf
// arms are not compiled at all (they were ~ 450 instructions per call site, skipped at run time, in the middle of the hot loop)
template <bool GEN = true> __device__ __forceinline__ float impedance(const float *solimp, float pos) {
    float dmin = fminf(fmaxf(solimp[0], HSR_MINIMP), HSR_MAXIMP), dmax = fminf(fmaxf(solimp[1], HSR_MINIMP), HSR_MAXIMP);
    const float width = fmaxf(solimp[2], HSR_MINVAL), mid = fminf(fmaxf(solimp[3], HSR_MINIMP), HSR_MAXIMP);
    const float power = fmaxf(solimp[4], 1.f);
    if (dmin == dmax || width <= HSR_MINVAL) return 0.5f * (dmin + dmax);
    const float x = fabsf(pos) * frcp(width);
    if (x >= 1) return dmax;
    if (x <= 0) return dmin;
    float y;
    if (power == 1.f) y = x;
    else if (!GEN || power == 2.f) y = x <= mid ? x * x * frcp(mid) : 1 - (1 - x) * (1 - x) * frcp(1 - mid);   // MuJoCo's default power: no powf
    else if (x <= mid) y = powf(x, power) / powf(mid, power - 1);
    else y = 1 - powf(1 - x, power) / powf(1 - mid, power - 1);
    return dmin + y * (dmax - dmin);
}
