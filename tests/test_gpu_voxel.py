"""GPU parity tests of the voxel front-end (lvba_voxmap_*) against oracle/voxel_oracle.py on identical scans.

Index work (root keys, octant paths, observing frames, admission) must agree exactly; the PointCluster sums are
accumulated in the reference's cloud order and must be BIT-identical; plane parameters (an eigen-decomposition) agree
to 1e-8 up to the sign of the normal."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [
    dict(n_frames=5, pts_per_frame=12000, room=(10, 8, 4), origin=(-3.3, 7.1, 0.4), n_panels=8, seed=11),
    dict(n_frames=3, pts_per_frame=9000, room=(6, 5, 3), origin=(40.2, -55.7, 3.0), n_panels=5, seed=12, voxel_size=0.5),
    dict(n_frames=8, pts_per_frame=5000, room=(10, 8, 4), origin=(0.0, 0.0, 0.0), n_panels=8, seed=13, voxel_size=2.0,
         point_floats=12),                                    # padded points (PCL PointXYZINormal stride)
]
STAGE2_RATIO = np.float32([0.08, 0.08, 0.08, 0.08])           # include/dataset_io.h:80


def _build(pkg, synth, case, ratio=None):
    from oracle import voxel_oracle as vo
    case = dict(case)
    vs = case.pop("voxel_size", 1.0)
    s = synth.make_scans(**case)
    ratio = vo.DEFAULT_EIGEN_RATIO if ratio is None else ratio
    surf_map, vox = vo.build([c[:, :3] for c in s["clouds"]], s["poses"], vs, ratio)
    m = pkg.VoxelMap(s["clouds"], s["poses"], vs, ratio)
    return s, vs, surf_map, vox, m


def _path_code(path):
    return len(path) | ((path[0] if len(path) >= 1 else 0) << 4) | ((path[1] if len(path) == 2 else 0) << 8)


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("ratio", [None, STAGE2_RATIO])
def test_voxels_match_oracle_bit_for_bit(pkg, synth, case, ratio):
    from oracle import voxel_oracle as vo
    s, vs, surf_map, vox, m = _build(pkg, synth, case, ratio)
    off_ref, idx_ref, cl_ref = vo.pack(vox)
    assert len(vox) > 20, "case too small to mean anything"
    assert m.info["n_points"] == sum(len(c) for c in s["clouds"])
    assert m.info["n_roots"] == len(surf_map)
    n_planes = 0

    def count(n):
        nonlocal n_planes
        n_planes += n.state == "PLANE"
        for l in n.leaves:
            if l is not None:
                count(l)
    for r in surf_map.values():
        count(r)
    assert m.info["n_planes"] == n_planes
    assert m.info["n_voxels"] == len(vox) and m.info["n_factors"] == off_ref[-1]
    off, idx, cl, key = m.export()
    key_ref = np.array([list(k) + [_path_code(p)] for k, p, _ in vox], np.int64)
    np.testing.assert_array_equal(key, key_ref)
    np.testing.assert_array_equal(off, off_ref)
    np.testing.assert_array_equal(idx, idx_ref)
    np.testing.assert_array_equal(cl, cl_ref)             # same products, same order of sums
    # the split branches were exercised
    layers = set(len(p) for _, p, _ in vox)
    assert layers >= {0, 1}


@pytest.mark.parametrize("case", CASES[:2])
def test_find_planes_matches_oracle(pkg, synth, case):
    from oracle import voxel_oracle as vo
    s, vs, surf_map, vox, m = _build(pkg, synth, case, STAGE2_RATIO)
    rng = np.random.default_rng(5)
    # queries: scan points moved to the world frame (mostly hits) + uniform points (mostly misses) + non-finite
    q = []
    for c, T in zip(s["clouds"], s["poses"]):
        sel = rng.choice(len(c), 300, replace=False)
        q.append(c[sel, :3].astype(np.float64) @ T[:9].reshape(3, 3).T + T[9:])
    org = np.asarray(case["origin"])
    q.append(org + rng.uniform(-6, 6, (500, 3)))
    q.append(np.array([[np.nan, 0, 0], [0, np.inf, 0]]))
    X = np.concatenate(q)
    plane, valid = m.find_planes(X)
    n_hit = 0
    for i, x in enumerate(X):
        ref = vo.find_plane(surf_map, x, vs)
        assert bool(valid[i]) == (ref is not None), i
        if ref is None:
            assert not plane[i].any()
            continue
        n_hit += 1
        n, d = ref
        sgn = np.sign(n @ plane[i, :3])
        # eigenvector of a (nearly) planar covariance: conditioning ~ eps * lambda_max / gap; d inherits it times |centre|
        assert np.abs(sgn * plane[i, :3] - n).max() < 1e-8
        assert abs(sgn * plane[i, 3] - d) < 1e-8 * (1 + np.abs(x).max())
    assert 200 < n_hit < len(X) - 200


def test_resident_scans_and_frame_windows(pkg, synth):
    """lvba_scans_create + lvba_voxmap_build_scans: clouds uploaded once, maps cut per window (the reference's
    runWindowBA loop, src/lvba_system.cpp:232-258) -- each must equal the oracle run on that window alone."""
    from oracle import voxel_oracle as vo
    s = synth.make_scans(9, 6000, room=(8, 6, 3), origin=(2.5, -1.5, 0.2), n_panels=6, seed=21, point_floats=5)
    with pkg.Scans(s["clouds"]) as scans:
        for begin, n in ((0, 9), (0, 4), (4, 5), (8, 1)):
            poses = s["poses"][begin:begin + n]
            surf_map, vox = vo.build([c[:, :3] for c in s["clouds"][begin:begin + n]], poses, 1.0)
            off_ref, idx_ref, cl_ref = vo.pack(vox)
            with scans.voxel_map(poses, 1.0, frame_begin=begin, n_frames=n) as m:
                assert m.info["n_points"] == sum(len(c) for c in s["clouds"][begin:begin + n])
                assert m.info["n_roots"] == len(surf_map) and m.info["n_voxels"] == len(vox)
                off, idx, cl, key = m.export()
                np.testing.assert_array_equal(off, off_ref)
                np.testing.assert_array_equal(idx, idx_ref)          # frame indices relative to the window
                np.testing.assert_array_equal(cl, cl_ref)
        L = __import__("importlib").import_module("global-lvba_amd._lib")
        with pytest.raises(L.LvbaError):
            scans.voxel_map(s["poses"][:3], 1.0, frame_begin=7, n_frames=3)


def test_map_feeds_the_lm_refinement(pkg, synth):
    """scans -> lvba_voxmap_to_balm -> damping_iter: the cost seen through the map equals the cost of the oracle's
    packed voxels, and refinement from odometry-grade poses reduces it."""
    from oracle import voxel_oracle as vo
    from oracle import balm_oracle as bo
    s, vs, surf_map, vox, m = _build(pkg, synth, CASES[0])
    off, idx, cl = vo.pack(vox)
    prob = m.tras_opt()
    c_gpu = prob.cost(s["poses"])
    c_ref = bo.only_residual(bo.Problem(len(s["poses"]), off, idx, cl), s["poses"])
    assert abs(c_gpu - c_ref) <= 1e-9 * c_ref
    poses, trace, rc = prob.refine(s["poses"])
    assert rc == 0
    assert trace[-1]["residual1"] <= trace[0]["residual1"]
    assert any(t["accepted"] for t in trace)


def test_argument_errors(pkg):
    from importlib import import_module
    L = import_module("global-lvba_amd._lib")
    T = np.concatenate([np.eye(3).reshape(-1), np.zeros(3)])[None]
    bad = np.array([[0.1, 0.2, np.nan]], np.float32)
    with pytest.raises(L.LvbaError) as e:
        pkg.VoxelMap([bad], T)
    assert e.value.code == L.ERR_ARG
    with pytest.raises(L.LvbaError):
        pkg.VoxelMap([np.zeros((4, 3), np.float32)], T, voxel_size=0.0)
    # a map without admitted voxels builds, answers lookups with "no plane", and refuses to become a problem
    m = pkg.VoxelMap([np.zeros((4, 3), np.float32)], T)
    assert m.info["n_voxels"] == 0 and m.info["n_roots"] == 1
    plane, valid = m.find_planes(np.zeros((3, 3)))
    assert not valid.any()
    with pytest.raises(L.LvbaError):
        m.tras_opt()
    empty = pkg.VoxelMap([np.zeros((0, 3), np.float32)], T)
    assert empty.info["n_points"] == 0
    assert not empty.find_planes(np.zeros((2, 3)))[1].any()


def test_cpp_adapter_runs_on_the_gpu(tmp_path):
    """include/lvba_adapter.hpp end to end from C++ (stand-in PCL/Eigen types): VoxelMap build, plane lookup,
    damping_iter -- the binding INTEGRATION.md shows, executed on the device."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "global-lvba_amd")
    exe = str(tmp_path / "adapter_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", os.path.join(root, "tests", "adapter_check.cpp"), "-o", exe,
                           "-L", libdir, "-llvba_hip", f"-Wl,-rpath,{libdir}"])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "voxel map on the GPU" in out.stdout and "refined on the GPU" in out.stdout and "cameras refined on the GPU" in out.stdout


def test_four_million_points_bit_identical_and_deterministic(pkg, synth):
    """At scale (16 scans x 250 k points, the bench's scene): the whole CSR output equals the C++ restatement bit for bit,
    two builds give identical bytes (no atomics on the data path), and every point is accounted for:
    sum of emitted cluster counts <= points, with equality once dropped / single-observer nodes are added back."""
    import oracle
    base = synth.make_scans(8, 250_000, room=(60, 40, 8), n_panels=16, n_blobs=40, origin=(120.0, -80.0, 2.0), point_floats=12)
    clouds, poses = [], []
    for r in range(2):
        for c, T in zip(base["clouds"], base["poses"]):
            T = T.copy(); T[9] += 100.0 * r
            clouds.append(c); poses.append(T)
    poses = np.asarray(poses)
    ref = oracle.voxel_build_cpp([c[:, :3] for c in clouds], poses, 0.5)
    with pkg.Scans(clouds) as scans:
        with scans.voxel_map(poses, 0.5) as m1:
            out1 = m1.export()
            info = dict(m1.info)
        with scans.voxel_map(poses, 0.5) as m2:
            out2 = m2.export()
    for a, b in zip(out1, out2):
        np.testing.assert_array_equal(a, b)
    off, idx, cl, key = out1
    assert info["n_points"] == 4_000_000 and info["n_roots"] == ref["n_roots"] and info["n_planes"] == ref["n_planes"]
    np.testing.assert_array_equal(off, ref["off"])
    np.testing.assert_array_equal(idx, ref["idx"])
    np.testing.assert_array_equal(key, ref["key"])
    np.testing.assert_array_equal(cl, ref["clu"])
    assert cl[:, 9].sum() <= info["n_points"] and (cl[:, 9] >= 1).all() and (np.diff(off) >= 2).all()
    assert len(off) - 1 > 50_000


def test_golden_fixture(pkg):
    """The HIP path against the committed fixture tests/golden/voxel_small.npz (inputs + frozen oracle answers): voxels and
    clusters bit for bit, plane lookups to 1e-8, the window stage's anchors and relative poses."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "voxel_small.npz"))
    clouds = np.split(z["points"], np.cumsum(z["counts"])[:-1])
    with pkg.Scans(clouds) as scans:
        with scans.voxel_map(z["poses"], 1.0) as m:
            off, idx, cl, key = m.export()
            plane, valid = m.find_planes(z["query"])
        w = scans.window_ba(z["poses"], window_size=3, voxel_size=1.0, eigen_ratio_array=np.float32([0.3, 0.1, 0.06, 0.03]),
                            anchor_leaf=0.05)
    np.testing.assert_array_equal(off, z["voxel_off"])
    np.testing.assert_array_equal(idx, z["pose_idx"])
    np.testing.assert_array_equal(key, z["voxel_key"])
    np.testing.assert_array_equal(cl, z["clusters"])
    np.testing.assert_array_equal(valid, z["valid"])
    sgn = np.sign(np.sum(plane[:, :3] * z["planes"][:, :3], axis=1))[:, None]
    assert np.abs(sgn * plane - z["planes"])[z["valid"] > 0].max() < 1e-8
    np.testing.assert_array_equal(w["anchor_index"], z["win_anchor_index"])
    assert np.abs(w["rel_poses"] - z["win_rel_poses"]).max() < 1e-7
    np.testing.assert_array_equal(w["anchor_poses"], z["win_anchor_poses"])
    got = w["anchor_scans"].counts.tolist()
    assert all(abs(a - b) <= max(2, 0.002 * b) for a, b in zip(got, z["win_anchor_counts"].tolist()))
    w["anchor_scans"].close()


def test_hip_matches_reference_voxel_golden(pkg):
    """The HIP voxel front-end against what the REFERENCE'S OWN cut_voxel / recut / tras_opt and findCorrespondPoint answer
    on tests/golden/voxel_small.npz's scans (tests/golden/ref_voxel.npz, generated by make_golden.py:main_ref from
    include/BALM/bavoxel.hpp compiled against the stand-ins of oracle/shim): same admitted voxels at the same octant paths,
    per-frame clusters bit for bit, plane lookups to 1e-8 up to the normal's sign."""
    import os
    gd = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    z, r = np.load(os.path.join(gd, "voxel_small.npz")), np.load(os.path.join(gd, "ref_voxel.npz"))
    clouds = np.split(z["points"], np.cumsum(z["counts"])[:-1])
    with pkg.Scans(clouds) as scans:
        with scans.voxel_map(z["poses"], 1.0) as m:
            off, idx, cl, key = m.export()
            plane, valid = m.find_planes(z["query"])
            info = dict(m.info)
    assert len(off) - 1 == int(r["n_admitted"]) == len(r["keys"])
    assert info["n_roots"] == int(r["n_roots"]) and info["n_planes"] == int(r["n_plane_nodes"])
    # our key column 3 is len | o1 << 4 | o2 << 8; the reference fixture's is layer << 6 | o1 << 3 | o2
    k3 = key[:, 3]
    ours = np.stack([key[:, 0], key[:, 1], key[:, 2], ((k3 & 15) << 6) | (((k3 >> 4) & 15) << 3) | ((k3 >> 8) & 15)], 1)
    order = np.lexsort((ours[:, 3], ours[:, 2], ours[:, 1], ours[:, 0]))
    np.testing.assert_array_equal(ours[order], r["keys"])
    slots = np.zeros_like(r["slots"])
    for a in range(len(off) - 1):
        for f in range(off[a], off[a + 1]):
            slots[a, idx[f]] = cl[f]
    np.testing.assert_array_equal(slots[order], r["slots"])
    want_valid = np.any(r["planes"] != 0, axis=1)
    np.testing.assert_array_equal(valid.astype(bool), want_valid)
    sgn = np.sign(np.sum(plane[:, :3] * r["planes"][:, :3], axis=1))[:, None]
    assert np.abs(sgn * plane - r["planes"])[want_valid].max() < 1e-8


def test_strided_scans_upload_packed(pkg, synth):
    """lvba_scans_create from a 48-byte point stride: the packed, pinned, multi-threaded upload puts the same xyz on the device
    as a 12-byte-stride upload of the coordinates alone -- frames longer and shorter than a 1 M-point chunk, an empty frame."""
    rng = np.random.default_rng(5)
    sizes = [3, 0, 70_000, 1_300_000, 1 << 20]
    clouds = [rng.standard_normal((n, 12)).astype(np.float32) for n in sizes]
    want = [np.ascontiguousarray(c[:, :3]) for c in clouds]
    with pkg.Scans(clouds) as sc:
        for f, w in enumerate(want):
            np.testing.assert_array_equal(sc.download(f), w)
    with pkg.Scans(want) as sc:
        for f, w in enumerate(want):
            np.testing.assert_array_equal(sc.download(f), w)


def test_wide_map_sorts_on_64_bit_repacked_keys(pkg, synth):
    """The root sort runs on the key bits that vary (csrc/voxel_internal.h: KeyPack): 32-bit keys when the three components'
    widths sum to <= 32 -- every other test here --, re-packed 64-bit keys otherwise.  Two rooms 150 km apart, on both sides of
    the origin, at 0.25 m voxels need 20 + 19 + 12 bits: same roots in the same order, same voxels, bit-identical clusters."""
    from oracle import voxel_oracle as vo
    a = synth.make_scans(3, 6000, room=(6, 5, 3), origin=(-70000.3, 61000.7, -250.4), n_panels=5, seed=21)
    b = synth.make_scans(3, 6000, room=(6, 5, 3), origin=(81000.2, -52000.9, 310.6), n_panels=5, seed=22)
    clouds = a["clouds"] + b["clouds"]
    poses = np.concatenate([a["poses"], b["poses"]])
    surf_map, vox = vo.build([c[:, :3] for c in clouds], poses, 0.25, vo.DEFAULT_EIGEN_RATIO)
    off_ref, idx_ref, cl_ref = vo.pack(vox)
    keys = np.array([k for k in surf_map.keys()], np.int64)
    widths = [int(np.ptp(keys[:, j])).bit_length() for j in range(3)]
    assert sum(widths) > 32, widths
    with pkg.VoxelMap(clouds, poses, 0.25, vo.DEFAULT_EIGEN_RATIO) as m:
        assert m.info["n_roots"] == len(surf_map)
        off, idx, cl, key = m.export()
    key_ref = np.array([list(k) + [_path_code(p)] for k, p, _ in vox], np.int64)
    assert len(vox) > 20
    np.testing.assert_array_equal(key, key_ref)
    np.testing.assert_array_equal(off, off_ref)
    np.testing.assert_array_equal(idx, idx_ref)
    np.testing.assert_array_equal(cl, cl_ref)


_SORT_SCRIPT = r"""
import importlib, sys, numpy as np
sys.path.insert(0, sys.argv[1])
pkg = importlib.import_module("global-lvba_amd")
synth = importlib.import_module("global-lvba_amd.synth")
s = synth.make_scans(8, 20000, room=(20, 14, 5), origin=(-12.5, 33.1, 1.2), n_panels=8, seed=31)
clouds = list(s["clouds"])
if sys.argv[3] == "skip":                 # the middle window sees volume noise only: no plane voxels, no anchor
    rng = np.random.default_rng(4)
    for f in (3, 4, 5):
        clouds[f] = (rng.uniform(-1, 1, (500, 3)) + np.array([-12.5, 33.1, 1.2])).astype(np.float32)
with pkg.Scans(clouds) as scans:
    with scans.voxel_map(s["poses"], 0.5) as m:
        off, idx, cl, key = m.export()
    w = scans.window_ba(s["poses"], window_size=3, voxel_size=0.5, anchor_leaf=0.05)       # 3 + 3 + 2 frames
    anchors = [w["anchor_scans"].download(k) for k in range(w["anchor_scans"].n_frames)]
    w["anchor_scans"].close()
np.savez(sys.argv[2], off=off, idx=idx, cl=cl, key=key, wp=w["window_poses"], n=np.array([len(a) for a in anchors]),
         pts=np.concatenate(anchors), rel=w["rel_poses"], ai=w["anchor_index"],
         skipped=np.array([x["skipped"] for x in w["windows"]]))
"""


@pytest.mark.parametrize("mode", ["plain", "skip"])
def test_joint_window_map_changes_no_byte(tmp_path, mode):
    """LVBA_WINDOW_JOINT_MAP=0 builds one voxel map per window (what a scan set too large for one joint map falls back to); the
    default builds ONE map whose roots are (window, key) and hands the windows views into it, and merges + down-samples the anchor
    clouds of all windows in one pass sorted by (window, leaf key) (window_ba.hip: stage_finish_joint).  Same order, same sums either way:
    the map and the whole window stage (refined window poses, anchor clouds point for point) are identical byte for byte."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "run.py"
    script.write_text(_SORT_SCRIPT)
    out = []
    for i, v in enumerate([{}, {"LVBA_WINDOW_JOINT_MAP": "0"}]):
        f = tmp_path / f"o_{i}.npz"
        r = subprocess.run([sys.executable, str(script), root, str(f), mode], env=dict(os.environ, **v), capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0, (v, r.stderr[-2000:])
        out.append(np.load(f))
    assert len(out[0]["off"]) > 100 and out[0]["n"].sum() > 1000
    assert out[0]["skipped"].tolist() == ([0, 1, 0] if mode == "skip" else [0, 0, 0])   # (skipped windows' points sort behind the others')
    for o in out[1:]:
        for k in out[0].files:
            np.testing.assert_array_equal(out[0][k], o[k], err_msg=k)
