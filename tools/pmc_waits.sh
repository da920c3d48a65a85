#!/bin/bash
# where the waves of the persistent kernel wait: tools/pmc_waits.sh CFG   (three PMC passes, per wave and substep)
CFG=${1:-cfg3}; WAVES=${2:-2048}
OUT=gpurun_out/pmc_wait_$CFG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH -d $OUT/a -- python3 bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-capacity > $OUT/a.log 2>&1 || exit 1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU -d $OUT/b -- python3 bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-capacity > $OUT/b.log 2>&1 || exit 1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_FLAT -d $OUT/c -- python3 bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-capacity > $OUT/c.log 2>&1
python3 - <<PY
import csv, glob
from collections import defaultdict
for d in ('a', 'b', 'c'):
    acc = defaultdict(float); n = set()
    for f in glob.glob('$OUT/' + d + '/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if 'k_env_step_mf' in r['Kernel_Name']:
                acc[r['Counter_Name']] += float(r['Counter_Value']); n.add(r['Dispatch_Id'])
    print(d, 'launches', len(n), 'per wave and substep:', {k: round(v / max(len(n), 1) / $WAVES / 300, 1) for k, v in acc.items()})
PY
