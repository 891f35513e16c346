#!/bin/bash
# front-end check: voxel / window / pipeline / fusion tests (bit-identical clusters), then the bench's front-end leg
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fe; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_voxel.py tests/test_gpu_window.py tests/test_gpu_pipeline.py -x -q -p no:cacheprovider 2>&1 | tail -4
python - <<'PY'
import importlib, json, sys
sys.argv = ["bench.py"]
import bench
pkg = importlib.import_module("global-lvba_amd"); synth = importlib.import_module("global-lvba_amd.synth")
fe = bench.front_end_leg(pkg, synth, False)
print(json.dumps({k: fe[k] for k in ("map_ms", "phase_ms", "points_per_s")}), fe["window_ba"]["ms_per_window"], fe["window_ba"]["stage_ms_per_window"], fe["roofline"]["frac"])
PY
