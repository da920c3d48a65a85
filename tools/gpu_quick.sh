#!/bin/bash
# quick GPU check: selected tests + bench line.  usage: tools/gpu_quick.sh TAG [pytest -k expr]
TAG=${1:-q}; K=${2:-}
mkdir -p gpurun_out/r3
if [ -n "$K" ]; then python -m pytest tests -m gpu -x -q -k "$K" > gpurun_out/r3/${TAG}_tests.log 2>&1; else python -m pytest tests -m gpu -x -q > gpurun_out/r3/${TAG}_tests.log 2>&1; fi
echo "tests rc=$?" >> gpurun_out/r3/${TAG}_tests.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-capacity > gpurun_out/r3/${TAG}_bench.log 2>&1 || exit 1
