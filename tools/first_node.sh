#!/bin/bash
# First contact with a multi-GPU node (no round had one): ONE command.
#   bash tools/first_node.sh [N=8]
# 1. tools/rccl_smoke.py on N ranks: RCCL carries the product's all-reduces for the first time -- ranks bitwise equal, result held
#    against the host-staged transport of the multi-rank tests, bus bandwidth of the pose-block all-reduce measured and written to
#    profiles/rccl_bus_bandwidth.json (bench.py's scaling_model then uses it in place of the assumed 250 GB/s);
# 2. bench.py --gpus 1 / 2 / 4 / 8 for C3 and C4 -> profiles/first_node_<config>_<n>gpu.json (one JSON line each).
set -u
N=${1:-8}
R=$(cd "$(dirname "$0")/.." && pwd); cd "$R"
export HSA_ENABLE_IPC_MODE_LEGACY=0
python -c "import __graft_entry__ as g; g.build()" || exit 1
python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port 29533 tools/rccl_smoke.py || echo "rccl_smoke FAILED (see profiles/rccl_smoke_${N}gpu.json)"
for cfg in C3 C4; do
  for n in 1 2 4 8; do
    [ "$n" -le "$N" ] || continue
    extra="--no-visual --no-front-end --no-y32 --no-reference-baseline"
    [ "$n" -gt 1 ] && extra="$extra --no-cpu-baseline"
    [ "$cfg" = C4 ] && extra="$extra --steps 10 --warmup 2 --no-cpu-baseline"
    if [ "$n" -eq 1 ]; then python bench.py --gpus 1 --config $cfg $extra > profiles/first_node_${cfg}_${n}gpu.json 2> profiles/first_node_${cfg}_${n}gpu.err
    else python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29540 + n)) bench.py --gpus $n --config $cfg $extra > profiles/first_node_${cfg}_${n}gpu.json 2> profiles/first_node_${cfg}_${n}gpu.err; fi
    echo "$cfg on $n GPU(s): $(grep -o '"value": [0-9.]*' profiles/first_node_${cfg}_${n}gpu.json | head -1) $(grep -o '"ms_per_step": [0-9.]*' profiles/first_node_${cfg}_${n}gpu.json | head -1)"
  done
done
