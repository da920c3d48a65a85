#!/bin/bash
# longer A/B of build variants: bench.py --steps 60 per variant (a change of rounding changes which envs become hard from the third env-step
# on, so ten steps are one draw of the slowest env, not a measurement): tools/ab_long.sh NAME...
for v in "$@"; do
  HSR_LIB=$PWD/hsr_env_amd/var_${v}.so python bench.py --steps ${AB_STEPS:-60} --warmup 3 --no-cpu-baseline --no-capacity --config ${HSR_CFG:-cfg3} 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('%-12s %7.1fk  kernel mean %.2f min %.2f max %.2f' % ('$v', d['value'] / 1e3, r['kernel_ms_mean'], r['kernel_ms_min'], r['kernel_ms_max']))"
done
