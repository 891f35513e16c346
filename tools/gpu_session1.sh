#!/bin/bash
# first GPU session: parity tests, probe timings, rocprof kernel stats
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
timeout 600 python tools/gpu_probe.py 100x20000 C2 > gpurun_out/probe_c2.log 2>&1
echo "probe exit $?" >> gpurun_out/probe_c2.log
tail -5 gpurun_out/pytest_gpu.log; tail -30 gpurun_out/probe_c2.log
