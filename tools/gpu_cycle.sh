#!/bin/bash
# one measurement cycle on the GPU box: parity tests, the bench line and the per-workgroup / per-phase profile
# usage: tools/gpu_cycle.sh TAG [pytest-args]
TAG=${1:-x}; shift
mkdir -p gpurun_out/r6
python -m pytest tests -m gpu -x -q "$@" > gpurun_out/r6/${TAG}_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r6/${TAG}_tests.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-capacity > gpurun_out/r6/${TAG}_bench.log 2>&1 || exit 1
HSR_LIB=hsr_env_amd/libhsrsim_timing.so python tools/block_times.py > gpurun_out/r6/${TAG}_bt.log 2>&1
HSR_LIB=hsr_env_amd/libhsrsim_life.so python tools/block_life.py > gpurun_out/r6/${TAG}_life.log 2>&1
python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline --no-capacity > gpurun_out/r6/${TAG}_bench4.log 2>&1 || exit 1
HSR_CFG=cfg4 HSR_LIB=hsr_env_amd/libhsrsim_timing.so python tools/block_times.py > gpurun_out/r6/${TAG}_bt4.log 2>&1
