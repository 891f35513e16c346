"""CPU tests of the ROS/PCL-free dataset I/O (global-lvba_amd/dataset.py): format known-answers written by hand from the
PCD v0.7 / TUM / COLMAP text specifications, and round trips."""
import importlib
import os

import numpy as np
import pytest

ds = importlib.import_module("global-lvba_amd.dataset")


def test_timestamp_from_name():
    assert ds.parse_timestamp_from_name("1699912345.123456.pcd") == 1699912345.123456
    assert ds.parse_timestamp_from_name("frame_12.5.png") == 12.5
    assert ds.parse_timestamp_from_name("scan42.pcd") == 42.0
    assert ds.parse_timestamp_from_name("nodigits.pcd") is None


def test_tum_known_answer_and_stride(tmp_path):
    p = tmp_path / "poses.txt"
    p.write_text("# timestamp tx ty tz qx qy qz qw\n"
                 "1.0 1 2 3 0 0 0 1\n"
                 "\n"
                 "2.0 0 0 0 0 0 0.7071067811865476 0.7071067811865476\n"      # +90 deg about z
                 "garbage line\n"
                 "3.0 4 5 6 0 0 0 2\n")                                         # un-normalised quaternion
    ts, poses = ds.load_poses_tum(str(p))
    assert ts.tolist() == [1.0, 2.0, 3.0]
    np.testing.assert_allclose(poses[0], [1, 0, 0, 0, 1, 0, 0, 0, 1, 1, 2, 3], atol=1e-15)
    np.testing.assert_allclose(poses[1][:9].reshape(3, 3), [[0, -1, 0], [1, 0, 0], [0, 0, 1]], atol=1e-15)
    np.testing.assert_allclose(poses[2][:9].reshape(3, 3), np.eye(3), atol=1e-15)   # normalised on load
    ts2, poses2 = ds.load_poses_tum(str(p), stride=2)
    assert ts2.tolist() == [1.0, 3.0]
    out = tmp_path / "out.txt"
    ds.write_poses_tum(str(out), ts, poses)
    _, back = ds.load_poses_tum(str(out))
    np.testing.assert_allclose(back, poses, atol=1e-8)


def test_pcd_ascii_known_answer(tmp_path):
    p = tmp_path / "a.pcd"
    p.write_text("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\n"
                 "TYPE F F F F\nCOUNT 1 1 1 1\nWIDTH 2\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS 2\nDATA ascii\n"
                 "1.5 -2.25 3.0 10\n0.1 0.2 0.3 255\n")
    pts, names = ds.load_pcd(str(p))
    assert names == ["x", "y", "z", "intensity"]
    np.testing.assert_array_equal(pts, np.float32([[1.5, -2.25, 3.0, 10], [0.1, 0.2, 0.3, 255]]))


@pytest.mark.parametrize("mode", ["ascii", "binary", "binary_compressed"])
def test_pcd_round_trip(tmp_path, mode):
    rng = np.random.default_rng(3)
    pts = rng.normal(size=(257, 4)).astype(np.float32)
    p = str(tmp_path / f"c_{mode}.pcd")
    ds.save_pcd(p, pts, mode=mode)
    back, names = ds.load_pcd(p)
    np.testing.assert_array_equal(back, pts)
    xyz, names = ds.load_pcd(p, fields=("x", "y", "z"))
    assert names == ["x", "y", "z"] and xyz.shape == (257, 3)


def test_pcd_binary_with_extra_fields(tmp_path):
    """PointXYZINormal-like records (padding / normals / curvature) written by hand: only the asked fields come back."""
    n = 5
    dt = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("_", "<f4"), ("normal_x", "<f4"), ("intensity", "<f4"),
                   ("ring", "<u2"), ("pad", "<u2")])
    rec = np.zeros(n, dt)
    rec["x"], rec["y"], rec["z"], rec["intensity"], rec["ring"] = np.arange(n), 2 * np.arange(n), -np.arange(n), 7, 3
    p = tmp_path / "x.pcd"
    hdr = ("VERSION 0.7\nFIELDS x y z _ normal_x intensity ring pad\nSIZE 4 4 4 4 4 4 2 2\nTYPE F F F F F F U U\n"
           f"COUNT 1 1 1 1 1 1 1 1\nWIDTH {n}\nHEIGHT 1\nPOINTS {n}\nDATA binary\n")
    p.write_bytes(hdr.encode() + rec.tobytes())
    pts, names = ds.load_pcd(str(p))
    assert names == ["x", "y", "z", "intensity"]
    np.testing.assert_array_equal(pts[:, 1], 2 * np.arange(n, dtype=np.float32))
    assert (pts[:, 3] == 7).all()


def test_lzf_back_references():
    # hand-assembled stream: literal "abc", then a back reference of length 6 at distance 3 -> "abcabcabc"
    stream = bytes([2]) + b"abc" + bytes([(4 << 5) | 0, 2])
    assert ds.lzf_decompress(stream, 9) == b"abcabcabc"


def test_dataset_directory(tmp_path):
    d = tmp_path / "all_pcd_body"
    d.mkdir()
    rng = np.random.default_rng(0)
    clouds = [rng.normal(size=(10 + i, 4)).astype(np.float32) for i in range(3)]
    for t, c in zip((2.5, 0.5, 1.5), clouds):                     # unsorted on disk
        ds.save_pcd(str(d / f"{t}.pcd"), c)
    (d / "lidar_poses.txt").write_text("0.5 0 0 0 0 0 0 1\n1.5 1 0 0 0 0 0 1\n2.5 2 0 0 0 0 0 1\n")
    out = ds.load_dataset(str(tmp_path))
    assert out["timestamps"].tolist() == [0.5, 1.5, 2.5]
    assert [len(c) for c in out["clouds"]] == [11, 12, 10]
    assert out["poses"][:, 9].tolist() == [0, 1, 2]


def test_colmap_text_writers(tmp_path):
    ds.write_images_txt(str(tmp_path / "images.txt"), [[1, 0, 0, 0], [0.5, 0.5, 0.5, 0.5]], [[0, 0, 0], [1, 2, 3]])
    lines = (tmp_path / "images.txt").read_text().splitlines()
    assert lines[0] == "0 1.000000 0.000000 0.000000 0.000000 0.000000 0.000000 0.000000 1 0.jpg"
    assert lines[1] == "0.0 0.0 -1" and lines[2].endswith("1 1.jpg")
    ds.write_points3d_txt(str(tmp_path / "points3D.txt"), [[1, 2, 3.5]], [[255, 0, 7]])
    assert (tmp_path / "points3D.txt").read_text() == "0 1.000000 2.000000 3.500000 255 0 7 0\n"


def test_colmap_database(tmp_path):
    """A three-image COLMAP database written by hand: blobs as COLMAP stores them (float32 keypoints [rows, cols],
    uint32 match pairs keyed by pair_id = id_small * (2^31 - 1) + id_large)."""
    import sqlite3
    p = str(tmp_path / "db.db")
    con = sqlite3.connect(p)
    con.execute("CREATE TABLE images (image_id INTEGER PRIMARY KEY, name TEXT)")
    con.execute("CREATE TABLE keypoints (image_id INTEGER PRIMARY KEY, rows INTEGER, cols INTEGER, data BLOB)")
    con.execute("CREATE TABLE two_view_geometries (pair_id INTEGER PRIMARY KEY, rows INTEGER, cols INTEGER, data BLOB)")
    names = {7: "0.5.jpg", 3: "1.5.jpg", 9: "2.5.jpg"}
    rng = np.random.default_rng(0)
    kp = {iid: rng.uniform(0, 600, (5 + iid, 6 if iid == 9 else 4)).astype(np.float32) for iid in names}
    for iid, n in names.items():
        con.execute("INSERT INTO images VALUES (?, ?)", (iid, n))
        con.execute("INSERT INTO keypoints VALUES (?, ?, ?, ?)", (iid, kp[iid].shape[0], kp[iid].shape[1], kp[iid].tobytes()))
    m37 = np.array([[0, 1], [2, 3], [7, 11], [99, 0]], np.uint32)          # (image 3, image 7); last row out of range
    con.execute("INSERT INTO two_view_geometries VALUES (?, ?, ?, ?)", (ds.image_ids_to_pair_id(7, 3), 4, 2, m37.tobytes()))
    con.commit(); con.close()
    assert ds.image_ids_to_pair_id(7, 3) == 3 * 2147483647 + 7 == ds.image_ids_to_pair_id(3, 7)
    order = ["0.5.jpg", "1.5.jpg", "2.5.jpg", "missing.jpg"]              # caller order: ids 7, 3, 9, none
    kps, matches = ds.load_colmap_db(p, order, [(0, 1), (1, 0), (0, 2), (0, 3)])
    np.testing.assert_array_equal(kps[0], kp[7]); np.testing.assert_array_equal(kps[2], kp[9])
    assert kps[3].shape[0] == 0
    # pair (0, 1) = ids (7, 3): stored as (3, 7) -> columns swapped back to (image 7, image 3)
    np.testing.assert_array_equal(matches[0], [[1, 0], [3, 2], [11, 7]])
    np.testing.assert_array_equal(matches[1], [[0, 1], [2, 3], [7, 11]])
    assert len(matches[2]) == 0 and len(matches[3]) == 0
