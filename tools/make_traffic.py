"""profiles/traffic_rNN.json from the two separate rocprofv3 PMC passes of tools/gpu_pmc2.sh (FETCH_SIZE, WRITE_SIZE):
HBM bytes per launch of the kernels of one H/g/cost evaluation and of the cost-only pass.

  python tools/make_traffic.py <pmc_FETCH_SIZE.csv> <pmc_WRITE_SIZE.csv> <out.json> [config] [git hash]

The file is bound to the configuration it was taken on and to a hash of the kernel sources (bench.py: kernel_source_sha16):
bench.py quotes it only for that configuration, on one GPU, while those sources are unchanged.

Corrections (MI355X_MICROARCH.md, HBM / rocprofv3): the counters are in KiB; on gfx950 FETCH_SIZE = TCC_EA0_RDREQ x 64 B
tallies 128-byte requests at 64 B, so it is doubled -- calibrated here on balm_cost_kernel, whose algorithmic byte count is
known (84 B/factor); the gather kernels fill whole 128-byte L2 lines too, so the factor applies to them as well (an upper
bound for any 64-byte requests among them).  WRITE_SIZE x1.  bench.py only reports a `traffic` whose kernel list is the one
it times (EVAL_KERNELS / COST_KERNELS there)."""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

EVAL_KERNELS = ["balm_voxel_kernel", "balm_factor_kernel", "balm_diag_reduce_kernel", "balm_pair_col_kernel", "balm_pair_reduce_kernel"]
COST_KERNELS = ["balm_voxel_kernel"]   # the LM loop costs its trial point with the voxel pass (cost + voxel records)


def read(path):
    out = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            if "<true>" in row["Name"].split("(")[0]:
                continue                                                     # the fp32-record variants of a run that also timed LVBA_Y32=1
            name = row["Name"].split("(")[0].split("::")[-1].split("<")[0]   # (template arguments dropped)
            out[name] = dict(kib=float(row["MeanValue"]), calls=int(row["Dispatches"]), ns=float(row["MeanDurationNs"]))
    return out


def main():
    fetch, write = read(sys.argv[1]), read(sys.argv[2])
    per = {}
    for k in dict.fromkeys(EVAL_KERNELS + COST_KERNELS):
        if k not in fetch or k not in write:
            raise SystemExit(f"kernel {k} is not in the PMC passes: were they taken from another build?")
        per[k] = {"fetch": 2.0 * 1024.0 * fetch[k]["kib"], "write": 1024.0 * write[k]["kib"], "raw_fetch_kib": fetch[k]["kib"],
                  "raw_write_kib": write[k]["kib"], "avg_ms_under_pmc": 1e-6 * fetch[k]["ns"]}
    out = {
        "eval": sum(per[k]["fetch"] + per[k]["write"] for k in EVAL_KERNELS),
        "cost": sum(per[k]["fetch"] + per[k]["write"] for k in COST_KERNELS),
        "eval_kernels": EVAL_KERNELS, "cost_kernels": COST_KERNELS,
        "unit": "bytes per launch (1 GPU)",
        "config": sys.argv[4] if len(sys.argv) > 4 else "C3",
        "kernel_source_sha16": __import__("bench").kernel_source_sha16(),
        "git": sys.argv[5] if len(sys.argv) > 5 else None,
        "source": "tools/gpu_pmc2.sh: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over bench.py --no-visual --no-front-end",
        "correction": "KiB -> bytes; FETCH_SIZE x2 (gfx950: 128-byte requests tallied at 64 B; calibrated on balm_cost_kernel), WRITE_SIZE x1",
        "per_kernel": per,
    }
    with open(sys.argv[3], "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: out[k] for k in ("eval", "cost")}))


if __name__ == "__main__":
    main()
