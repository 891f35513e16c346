#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/bench.log 2>gpurun_out/bench.err
echo "bench exit $?"; tail -3 gpurun_out/bench.err; cat gpurun_out/bench.log
