#!/bin/bash
# instruction mix of the persistent kernel (two more PMC passes, each after three warm-up env-steps; tools/pmc_collect.py reads the last launch):
# branches, scalar / vector memory instructions, MFMA, transcendental ops, VALU instructions skipped with EXEC = 0, cycles waiting on LDS; and the
# cycles a wave spends issuing each instruction class.  usage: tools/pmc_mix.sh TAG
TAG=${1:-rXX}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VSKIPPED SQ_WAIT_INST_LDS -d $OUT/pmc_mixa -- python3 bench.py --steps 1 --warmup 3 --no-cpu-baseline --no-capacity > $OUT/pmc_mixa.log 2>&1 || exit 1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES -d $OUT/pmc_mixb -- python3 bench.py --steps 1 --warmup 3 --no-cpu-baseline --no-capacity > $OUT/pmc_mixb.log 2>&1 || exit 1
echo done
