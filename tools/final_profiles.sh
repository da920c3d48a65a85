#!/bin/bash
# Evidence of a round on the GPU box in one call: rocprofv3 kernel-trace stats + PMC passes (tools/profile_round.sh), workgroup lifetimes of the
# product kernel for the five configurations (libhsrsim_life.so), the phase profile of cfg3 / cfg4 (libhsrsim_timing.so).  usage: tools/final_profiles.sh TAG
TAG=${1:-r06}
R=${HSR_ROUND_DIR:-gpurun_out/r6}; mkdir -p $R
export TMPDIR=/tmp
bash tools/profile_round.sh $TAG > $R/final_profile_round.log 2>&1 || { echo "profile_round failed"; tail -5 $R/final_profile_round.log; exit 1; }
bash tools/pmc_mix.sh $TAG > $R/final_pmc_mix.log 2>&1 || { echo "pmc_mix failed"; exit 1; }
for c in cfg1 cfg2 cfg3 cfg4 cupboard; do HSR_CFG=$c HSR_LIB=$PWD/hsr_env_amd/libhsrsim_life.so python tools/block_life.py --json > $R/final_life_$c.log 2>&1 || exit 1; done
HSR_LIB=$PWD/hsr_env_amd/libhsrsim_timing.so python tools/block_times.py > $R/final_bt.log 2>&1 || exit 1
HSR_CFG=cfg4 HSR_LIB=$PWD/hsr_env_amd/libhsrsim_timing.so python tools/block_times.py > $R/final_bt4.log 2>&1 || exit 1
HSR_LIB=$PWD/hsr_env_amd/libhsrsim_life.so python tools/env_life.py 8192 cfg3 > $R/final_envlife.log 2>&1
echo done
