"""Where the time of a step launch of the band LDL^T goes: wall-clock marks of the role workgroups and of every bulk workgroup
(LVBA_STAMPS build, tools/build_variant.py stamps -DLVBA_STAMPS), for a few launches of the last solve of a short C3 run.
usage: LVBA_HIP_LIB=ab/liblvba_stamps.so python tools/step_stamps.py [config] [launch indices...]"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

pkg = importlib.import_module("global-lvba_amd")
synth = importlib.import_module("global-lvba_amd.synth")

NL, NR, NM, NB = 256, 96, 12, 640
CH = ["start", "loads there, staged", "product L", "put L, Z + stores", "product p", "W built", "diag factor", "end", "loads requested"]
RW = ["start", "q product(s)", "A, G staged", "product L", "product Z", "put + stores", "final product", "end", "loads requested", "first tiles staged"]
TICK_MHZ = 100.0  # s_memrealtime: calibrated against the host clock below (it differed between the boxes of round 6)
BT = ["start", "first fetches out", "chunk 0 staged", "products 0", "chunk 1 staged", "products 1", "chunk 2 staged", "products 2",
      "chunk 3 staged", "products 3", "end (C stored)"]


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
    picks = [int(a) for a in sys.argv[2:]] or [10, 11, 40, 41, 90, 91]
    N, V = bench.parse_config(cfg, synth)
    d = synth.make_balm_problem(N, V, device="cuda:0")
    prob = pkg.BalmProblem(N, d["voxel_off"], d["pose_idx"], d["clusters"], device=0)
    prob.refine(d["poses_init"])
    prob.refine(d["poses_init"])
    lib = pkg._lib.load()
    global TICK_MHZ
    import time
    lib.lvba_debug_tick.restype = C.c_uint64
    lib.lvba_debug_tick()
    h0, t0 = time.perf_counter(), lib.lvba_debug_tick()
    time.sleep(0.5)
    h1, t1 = time.perf_counter(), lib.lvba_debug_tick()
    TICK_MHZ = (t1 - t0) / (h1 - h0) / 1e6
    print(f"s_memrealtime: {TICK_MHZ:.1f} MHz on this box")
    lib.lvba_debug_bulk_marks.restype = C.c_int64
    lib.lvba_debug_bulk_marks.argtypes = [C.c_void_p, C.c_int64]
    lib.lvba_debug_stamps.restype = C.c_int64
    lib.lvba_debug_stamps.argtypes = [C.c_void_p, C.c_int64]
    n = NL * NR * NM + NL * NB * 2
    buf = np.zeros(n, np.uint64)
    got = lib.lvba_debug_stamps(buf.ctypes.data, n)
    assert got == n, got
    roles = buf[:NL * NR * NM].reshape(NL, NR, NM).astype(np.int64)
    bulk = buf[NL * NR * NM:].reshape(NL, NB, 2).astype(np.int64)
    nm = NL * NB * 12
    mbuf = np.zeros(nm, np.uint64)
    assert lib.lvba_debug_bulk_marks(mbuf.ctypes.data, nm) == nm
    marks = mbuf.reshape(NL, NB, 12).astype(np.int64)
    dump = os.environ.get("LVBA_STAMPS_DUMP") # a .npz with the raw arrays (ticks) for offline analysis
    if dump:
        np.savez_compressed(dump, roles=roles, bulk=bulk, marks=marks, tick_mhz=TICK_MHZ)
    for L in picks:
        r, b = roles[L], bulk[L]
        live = r[:, 0] > 0
        if not live.any():
            print(f"launch {L}: no stamps")
            continue
        bl = (b[:, 0] > 0) & (b[:, 1] >= b[:, 0])
        t0 = min(r[live, 0].min(), b[bl, 0].min() if bl.any() else 1 << 62)
        us = lambda x: (x - t0) / TICK_MHZ
        end_all = max(r[live][:, 7].max(), b[bl, 1].max() if bl.any() else 0)
        print(f"== launch {L}: {int(live.sum())} role workgroups, {int(bl.sum())} bulk workgroups stamped; first start -> last end {us(end_all):.1f} us")
        for prob_i in (0, 1):
            ch = r[prob_i * 48]
            if ch[0] > 0:
                print(f"  chain[{prob_i}]: " + " | ".join(f"{CH[m]} {us(ch[m]):.1f}" for m in (0, 8, 1, 2, 3, 4, 5, 6, 7) if ch[m] > 0))
            rows = r[prob_i * 48 + 1: prob_i * 48 + 48]
            rows = rows[rows[:, 0] > 0]
            if len(rows):
                for m in (0, 8, 9, 1, 2, 3, 4, 5, 6, 7):
                    col = rows[:, m]
                    col = col[col > 0]
                    if len(col):
                        print(f"  rows[{prob_i}] {RW[m]:>16}: min {us(col.min()):5.1f}  median {us(np.median(col)):5.1f}  max {us(col.max()):5.1f}   ({len(col)} rows)")
        if bl.any():
            st, en = us(b[bl, 0]), us(b[bl, 1])
            du = en - st
            print(f"  bulk: start min {st.min():.1f} median {np.median(st):.1f} max {st.max():.1f} | end median {np.median(en):.1f} max {en.max():.1f} | "
                  f"duration min {du.min():.1f} median {np.median(du):.1f} max {du.max():.1f}")
            mk = marks[L]
            ok = (mk[:, 0] > 0) & (mk[:, 10] > 0)
            if ok.any():
                d = np.diff(mk[:, :11], axis=1) / TICK_MHZ
                cols = [m for m in range(10) if (mk[ok, m + 1] > 0).all() and (mk[ok, m] > 0).all()]
                print("  bulk tile phases (median over %d tiles, us): " % int(ok.sum()) + " | ".join(f"{BT[m + 1]} {np.median(d[ok, m]):.2f}" for m in cols))
                tot = (mk[:, 10] - mk[:, 0]) / TICK_MHZ
                worst = np.argsort(-np.where(ok, tot, -1))[:4]
                for wv in worst:
                    print(f"    tile {wv} ({tot[wv]:.1f} us): " + " | ".join(f"{BT[m + 1]} {d[wv, m]:.2f}" for m in cols))
            idx = np.nonzero(bl)[0]
            order = np.argsort(-du)[:12]
            print("        slowest bulk workgroups (index: start -> end): " + ", ".join(f"{idx[o]}: {st[o]:.1f}->{en[o]:.1f}" for o in order))
            q = np.percentile(en, [10, 25, 50, 75, 90, 100])
            print("        end times, percentiles 10/25/50/75/90/100: " + " ".join(f"{v:.1f}" for v in q))
            late = st > np.median(st) + 5
            print(f"        workgroups starting > 5 us after the median start: {int(late.sum())}")


if __name__ == "__main__":
    main()
