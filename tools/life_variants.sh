#!/bin/bash
# workgroup lifetimes of several diagnostic builds: [HSR_CFG=cfg4] tools/life_variants.sh NAME... (hsr_env_amd/libhsrsim_life_NAME.so)
mkdir -p gpurun_out/r3
for v in "$@"; do
  HSR_LIB=hsr_env_amd/libhsrsim_life_$v.so python tools/block_life.py > gpurun_out/r3/life_${HSR_CFG:-cfg3}_$v.log 2>&1 || exit 1
done
