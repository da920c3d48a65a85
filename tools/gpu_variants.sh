#!/bin/bash
# A/B of build variants on the GPU box: for each NAME, hsr_env_amd/var_NAME.so (bench) and var_NAME_t.so (block times)
mkdir -p gpurun_out/r3
for v in "$@"; do
  HSR_LIB=hsr_env_amd/var_${v}.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-capacity > gpurun_out/r3/v_${v}_bench.log 2>&1 || exit 1
  HSR_LIB=hsr_env_amd/var_${v}_t.so python tools/block_times.py > gpurun_out/r3/v_${v}_bt.log 2>&1 || exit 1
done
