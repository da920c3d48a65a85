#!/bin/bash
# rocprofv3 evidence of one round on the GPU box: kernel-trace stats of the bench command, then PMC passes (separate runs, with
# --kernel-trace only: FETCH_SIZE, WRITE_SIZE, SQ counters) of a one-step bench after three warm-up env-steps (tools/pmc_collect.py reads the last launch of the persistent kernel).  Outputs under gpurun_out/prof_$1/ ;
# tools/pmc_collect.py condenses them into profiles/.
TAG=${1:-rXX}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-capacity > $OUT/trace.log 2>&1 || exit 1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -- python3 bench.py --steps 1 --warmup 3 --no-cpu-baseline --no-capacity > $OUT/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -- python3 bench.py --steps 1 --warmup 3 --no-cpu-baseline --no-capacity > $OUT/pmc_write.log 2>&1 || exit 1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES -d $OUT/pmc_sq -- python3 bench.py --steps 1 --warmup 3 --no-cpu-baseline --no-capacity > $OUT/pmc_sq.log 2>&1 || exit 1
# where the HBM-side writes come from: full 64 B lines (wave-wide rows: the scratch spills) against 32 B partial writes (words of the env-strided state
# arrays, contact records), and how many write requests reached L2 at all
rocprofv3 --kernel-trace --output-format csv --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_WRITE_sum -d $OUT/pmc_wr -- python3 bench.py --steps 1 --warmup 3 --no-cpu-baseline --no-capacity > $OUT/pmc_wr.log 2>&1 || exit 1
find $OUT -name "*.csv" | head -20
