"""Pins oracle/voxel_oracle.py (the restated cut_voxel / recut / tras_opt / findCorrespondPoint) with hand-built
cases whose answers follow from reading include/BALM/bavoxel.hpp:320-474,799-836 directly.  CPU only."""
import numpy as np
import pytest

from oracle import voxel_oracle as vo

I12 = np.concatenate([np.eye(3).reshape(-1), np.zeros(3)])


def test_root_key_quirks():
    # float quotient, "-1 if negative", then C truncation (bavoxel.hpp:809-815)
    assert vo.root_key(np.array([0.5, 1.5, 2.999]), 1.0) == (0, 1, 2)
    assert vo.root_key(np.array([-0.5, -1.0, -1.5]), 1.0) == (-1, -2, -2)       # exact negative integers shift down
    assert vo.root_key(np.array([0.99999999999, 3.0, -1e-12]), 1.0) == (1, 3, -1)  # fp32 rounding reaches the next voxel
    assert vo.root_key(np.array([1.0, 2.0, -3.0]), 0.5) == (2, 4, -7)


def _plane_patch(rng, n, center, normal, half=0.3, noise=0.0):
    normal = np.asarray(normal, float) / np.linalg.norm(normal)
    a = np.cross(normal, [0.3, 0.5, 0.8]); a /= np.linalg.norm(a)
    b = np.cross(normal, a)
    uv = rng.uniform(-half, half, (n, 2))
    return (np.asarray(center) + uv[:, :1] * a + uv[:, 1:] * b + rng.normal(0, noise, (n, 1)) * normal).astype(np.float32)


def test_single_plane_two_frames_matches_direct_sums():
    rng = np.random.default_rng(1)
    A = _plane_patch(rng, 20, [0.5, 0.5, 0.5], [0, 0, 1], noise=0.002)
    B = _plane_patch(rng, 17, [0.5, 0.5, 0.5], [0, 0, 1], noise=0.002)
    surf_map, vox = vo.build([A, B], np.stack([I12, I12]), 1.0)
    assert list(surf_map) == [(0, 0, 0)] and len(vox) == 1
    key, path, node = vox[0]
    assert path == () and node.state == "PLANE" and node.layer == 0
    np.testing.assert_array_equal(node.center, np.float32([0.5, 0.5, 0.5]))
    assert node.quater == np.float32(0.25)
    for f, pts in enumerate((A, B)):
        p = pts.astype(np.float64)
        want = np.zeros(10)
        for q in p:                                    # sequential, as PointCluster::push (tools.hpp:428-433)
            want[:6] += [q[0] * q[0], q[0] * q[1], q[0] * q[2], q[1] * q[1], q[1] * q[2], q[2] * q[2]]
            want[6:9] += q
            want[9] += 1
        np.testing.assert_array_equal(node.sig[f], want)
    off, idx, cl = vo.pack(vox)
    assert off.tolist() == [0, 2] and idx.tolist() == [0, 1]
    assert abs(abs(node.plane_normal[2]) - 1) < 1e-3


def test_admission_rules():
    rng = np.random.default_rng(2)
    few = _plane_patch(rng, 7, [0.5, 0.5, 0.5], [0, 0, 1])
    _, vox = vo.build([few, few], np.stack([I12, I12]), 1.0)               # 14 < min_ps -> dropped (:399-405)
    assert vox == []
    one = _plane_patch(rng, 40, [0.5, 0.5, 0.5], [0, 0, 1])
    empty = np.zeros((0, 3), np.float32)
    surf_map, vox = vo.build([one, empty], np.stack([I12, I12]), 1.0)      # PLANE but one observer (:45-54)
    assert vox == [] and surf_map[(0, 0, 0)].state == "PLANE"
    blob = rng.uniform(0.05, 0.95, (4000, 3)).astype(np.float32)            # volume: splits twice, then dropped
    surf_map, vox = vo.build([blob, blob], np.stack([I12, I12]), 1.0)
    root = surf_map[(0, 0, 0)]
    assert vox == [] and root.state == "SPLIT"
    assert all(l is None or l.state in ("SPLIT", "MID_NODE") for l in root.leaves)


def test_split_into_child_planes_and_lookup():
    rng = np.random.default_rng(3)
    # two parallel thin slabs in one root voxel: z = 0.2 (lower half) and z = 0.8 (upper half) -> root is not planar,
    # the layer-1 children (x,y quadrants x z halves) each hold one slab piece and are planes
    lo = np.column_stack([rng.uniform(0.02, 0.98, (400, 2)), np.full(400, 0.2) + rng.normal(0, 0.002, 400)])
    hi = np.column_stack([rng.uniform(0.02, 0.98, (400, 2)), np.full(400, 0.8) + rng.normal(0, 0.002, 400)])
    pts = np.concatenate([lo, hi]).astype(np.float32)
    T = np.stack([I12, I12])
    T[1, 9:] = [0.0, 0.0, 0.0]
    surf_map, vox = vo.build([pts[::2], pts[1::2]], T, 1.0)
    root = surf_map[(0, 0, 0)]
    assert root.state == "SPLIT"
    assert len(vox) == 8 and all(len(p) == 1 for _, p, _ in vox)
    assert [p[0] for _, p, _ in vox] == list(range(8))                     # octant order 4x + 2y + z (:373)
    for _, (o,), n in vox:
        b = [(o >> 2) & 1, (o >> 1) & 1, o & 1]
        np.testing.assert_array_equal(n.center, np.float32([0.25 + 0.5 * b[0], 0.25 + 0.5 * b[1], 0.25 + 0.5 * b[2]]))
        assert n.quater == np.float32(0.125) and n.layer == 1
    # lookup: a point above z = 0.5 lands in the upper child whose plane is z = 0.8
    got = vo.find_plane(surf_map, np.array([0.3, 0.7, 0.9]), 1.0)
    assert got is not None
    n, d = got
    assert abs(abs(n[2]) - 1) < 1e-3 and abs(abs(d) - 0.8) < 5e-3
    got = vo.find_plane(surf_map, np.array([0.3, 0.7, 0.1]), 1.0)
    assert abs(abs(got[1]) - 0.2) < 5e-3
    assert vo.find_plane(surf_map, np.array([5.3, 0.7, 0.1]), 1.0) is None          # no such root
    assert vo.find_plane(surf_map, np.array([np.nan, 0.7, 0.1]), 1.0) is None


def test_pose_is_applied_before_hashing():
    rng = np.random.default_rng(4)
    body = _plane_patch(rng, 30, [0.5, 0.5, 0.5], [1, 0, 0], noise=0.001)
    T = np.stack([I12, I12])
    T[:, 9:] = [-7.0, 3.0, 12.0]
    surf_map, vox = vo.build([body, body], T, 1.0)
    assert list(surf_map) == [(-7, 3, 12)]
    assert len(vox) == 1
    np.testing.assert_array_equal(vox[0][2].center, np.float32([-6.5, 3.5, 12.5]))
    # clusters stay in the BODY frame
    assert abs(vox[0][2].sig[0][6] / 30 - 0.5) < 0.05


@pytest.mark.parametrize("ratio", [vo.DEFAULT_EIGEN_RATIO, np.float32([0.08, 0.08, 0.08, 0.08])])
def test_cpp_twin_matches_bit_for_bit(ratio):
    """oracle/voxel_oracle.cpp (unordered_map + recursive octree, the reference's data structures) against the Python
    restatement on a synthetic window: keys, paths, frames and cluster sums identical."""
    import importlib
    import oracle
    synth = importlib.import_module("global-lvba_amd.synth")
    s = synth.make_scans(5, 12000, room=(10, 8, 4), origin=(-3.3, 7.1, 0.4), n_panels=8, seed=11)
    surf_map, vox = vo.build(s["clouds"], s["poses"], 1.0, ratio)
    off, idx, cl = vo.pack(vox)
    got = oracle.voxel_build_cpp(s["clouds"], s["poses"], 1.0, ratio)
    assert got["n_roots"] == len(surf_map) and len(vox) > 50
    key_ref = np.array([list(k) + [len(p) | ((p[0] if len(p) >= 1 else 0) << 4) | ((p[1] if len(p) == 2 else 0) << 8)]
                        for k, p, _ in vox], np.int64)
    np.testing.assert_array_equal(got["key"], key_ref)
    np.testing.assert_array_equal(got["off"], off)
    np.testing.assert_array_equal(got["idx"], idx)
    np.testing.assert_array_equal(got["clu"], cl)
    assert {len(p) for _, p, _ in vox} >= {0, 1}


def _golden():
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "voxel_small.npz"))
    clouds = np.split(z["points"], np.cumsum(z["counts"])[:-1])
    return z, clouds


def test_oracles_reproduce_golden_fixture():
    """tests/golden/voxel_small.npz freezes both restatements (and the window oracle built on them)."""
    import oracle
    from oracle import window_oracle as wo
    z, clouds = _golden()
    surf_map, vox = vo.build(clouds, z["poses"], 1.0)
    off, idx, cl = vo.pack(vox)
    np.testing.assert_array_equal(off, z["voxel_off"])
    np.testing.assert_array_equal(idx, z["pose_idx"])
    np.testing.assert_array_equal(cl, z["clusters"])
    cpp = oracle.voxel_build_cpp(clouds, z["poses"], 1.0)
    np.testing.assert_array_equal(cpp["clu"], z["clusters"])
    np.testing.assert_array_equal(cpp["key"], z["voxel_key"])
    for i, x in enumerate(z["query"]):
        r = vo.find_plane(surf_map, x, 1.0)
        assert (r is not None) == bool(z["valid"][i])
        if r is not None:
            assert np.abs(np.r_[r[0], r[1]] - z["planes"][i]).max() < 1e-12
    w = wo.run_window_ba(clouds, z["poses"], 3, 1.0, np.float32([0.3, 0.1, 0.06, 0.03]), 0.05)
    np.testing.assert_array_equal(w["anchor_index"], z["win_anchor_index"])
    assert np.abs(w["rel_poses"] - z["win_rel_poses"]).max() < 1e-10
    assert [len(c) for c in w["anchor_clouds"]] == z["win_anchor_counts"].tolist()
