"""CPU test of the host-side pose ordering (global-lvba_amd/csrc/ordering.h: reverse Cuthill-McKee from two start rules +
barycenter sweeps + hill-climbing), compiled with g++ through tests/ordering_check.cpp: the result is a permutation, never
wider than plain RCM or the natural order, and identical from call to call (every rank of a multi-GPU job must derive the
same order from the same all-reduced graph)."""
import os
import re
import subprocess

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("ordering") / "ordering_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "ordering_check.cpp"), "-o", out])
    return out


@pytest.mark.parametrize("n,band,loop", [(2000, 50, 50), (600, 20, 50), (300, 10, 0), (100, 5, 0), (64, 3, 100)])
def test_ordering_properties(exe, n, band, loop):
    r = subprocess.run([exe, str(n), str(band), str(loop)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"natural=(\d+) rcm=(\d+) ordered=(\d+)", r.stdout)
    nat, rcm, got = (int(v) for v in m.groups())
    assert got <= rcm and got <= nat
    if loop and n >= 600:            # ring trajectories with loop closures: the refinement beats plain RCM clearly
        assert got <= 0.8 * rcm


@pytest.mark.parametrize("shape,n,reach,ranks,kind", [("ring", 2000, 50, 1, "band"), ("lot", 2000, 50, 1, "hubs"),
                                                     ("long", 4000, 30, 4, "chunks"), ("long", 3000, 20, 8, "chunks")])
def test_nested_dissection_plan(tmp_path_factory, shape, n, reach, ranks, kind):
    """global-lvba_amd/csrc/nd_plan.h through tests/nd_plan_check.cpp: the folded ring of config C3 keeps its band; a hub on the
    ring (a place crossed eight times) is taken out as the separator; a long band on several ranks is cut into arcs by chunks
    of its band ordering.  The check program verifies the partition itself (permutation, no edge between two arcs, every arc's
    separator list complete, the bandwidths it reports, every rank owns an arc, determinism)."""
    exe = str(tmp_path_factory.mktemp("nd") / "nd_plan_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "nd_plan_check.cpp"), "-o", exe])
    r = subprocess.run([exe, shape, str(n), str(reach), str(ranks)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"kind={kind}" in r.stdout, r.stdout
    if kind != "band":
        assert "nd plan ok" in r.stdout


def test_host_tables(tmp_path):
    """global-lvba_amd/csrc/host_tables.h: chunks of the voxel-major kernels (<= 256 factors, <= 128 voxels, big voxels alone,
    greedy) and work items of the pair pass (lists longer than the cut become partial-block items), through
    tests/host_tables_check.cpp."""
    exe = str(tmp_path / "host_tables_check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", os.path.join(ROOT, "tests", "host_tables_check.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "host tables ok" in r.stdout, r.stdout + r.stderr


def test_host_arena(tmp_path):
    """global-lvba_amd/csrc/host_arena.h: the library-owned host blocks the set-up tables live in (size classes with <= 12.5 %
    slack, reuse from the class's free list, small blocks left to malloc, release, use from several threads), through
    tests/host_arena_check.cpp."""
    exe = str(tmp_path / "host_arena_check")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread", os.path.join(ROOT, "tests", "host_arena_check.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "host arena ok" in r.stdout, r.stdout + r.stderr


def test_ldlt_lookahead_schedule(tmp_path):
    """global-lvba_amd/csrc/ldlt_schedule.h: the launch list of the look-ahead band LDL^T (one launch per panel, no
    synchronisation inside a launch) replayed on a tile-level model of the factorisation -- one writer per tile and launch,
    inputs from earlier launches only, every contribution exactly once, nothing missing at the end; both schedules (single
    panels / paired rank-128 updates), both phase kinds (two-ended with an early close, top-down to the last panel), the
    geometries of configs C3 and C4 and a set of edge cases (tests/ldlt_schedule_check.cpp)."""
    exe = str(tmp_path / "ldlt_schedule_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "ldlt_schedule_check.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "ldlt schedule ok" in r.stdout, r.stdout[-3000:]
    # buffer lifetimes are part of the model (the ring of four Z buffers, the two side-copy / diagonal-share slots of ldlt.hip's
    # run_phase): with one Z slot fewer the same schedule must be refused
    exe3 = str(tmp_path / "ldlt_schedule_check_ring3")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-DLVBA_ZRING=3", os.path.join(ROOT, "tests", "ldlt_schedule_check.cpp"), "-o", exe3])
    r3 = subprocess.run([exe3], capture_output=True, text=True)
    assert r3.returncode != 0 and "ring slot" in r3.stdout, r3.stdout[-2000:]


def test_key_repacking_keeps_order_and_round_trips(tmp_path):
    """csrc/key_pack.h: the voxel map's root sort and the anchor down-sampling sort run on keys re-packed onto the bits that
    vary; the re-packed keys must order exactly like the 3 x 21-bit ones and expand back to them (tests/key_pack_check.cpp)."""
    exe = str(tmp_path / "key_pack_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "key_pack_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "key pack ok" in out.stdout, out.stdout[-2000:]
