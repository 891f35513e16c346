"""ROS/PCL-free readers and writers for the on-disk formats either side of the refinement path.

Mirrors (reference paths):
  parse_timestamp_from_name  include/utils.hpp:462-477        first "digits[.digits]" in a file name
  load_poses_tum             src/dataset_io.cpp:133-184       "t tx ty tz qx qy qz qw" per line, '#' comments, stride
  load_body_points           src/dataset_io.cpp:210-283       all_pcd_body/*.pcd sorted by time stamp, XYZI -> x, y, z, intensity
  load_pcd / save_pcd        PCD v0.7 (what pcl::io::loadPCDFile / savePCDFileBinary read and write): ascii, binary,
                             binary_compressed (LZF, fields stored one after the other)
  write_images_txt           src/lvba_system.cpp:2018-2024    COLMAP images.txt rows "id qw qx qy qz tx ty tz 1 id.jpg" + "0.0 0.0 -1"
  write_points3d_txt         src/lvba_system.cpp:2126-2137    COLMAP points3D.txt rows "i x y z r g b 0"
  load_colmap_db             src/lvba_system.cpp:510-685      keypoints + inlier matches from a COLMAP sqlite database
Host-side I/O only: nothing here touches the GPU; the arrays go straight into Scans / lidar_ba.
"""
from __future__ import annotations

import os
import re

import numpy as np

_TS_RE = re.compile(r"([0-9]+(?:\.[0-9]+)?)")


def parse_timestamp_from_name(fname):
    m = _TS_RE.search(os.path.basename(fname))
    return float(m.group(1)) if m else None


def quat_to_rot(qw, qx, qy, qz):
    """Eigen::Quaterniond(w, x, y, z).normalize() -> rotation matrix (src/dataset_io.cpp:166-167)."""
    n = np.sqrt(qw * qw + qx * qx + qy * qy + qz * qz)
    w, x, y, z = qw / n, qx / n, qy / n, qz / n
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def load_poses_tum(path, stride=1):
    """Returns (timestamps [n], poses [n, 12] = R row-major + t).  Unparsable lines are skipped, as upstream."""
    if stride < 1:
        raise ValueError("stride must be >= 1")
    ts, poses, valid = [], [], 0
    with open(path) as f:
        for line in f:
            if not line.strip() or line[0] == "#":
                continue
            tok = line.split()
            try:
                t, tx, ty, tz, qx, qy, qz, qw = (float(v) for v in tok[:8])
            except ValueError:
                continue
            if len(tok) < 8:
                continue
            if valid % stride == 0:
                ts.append(t)
                poses.append(np.concatenate([quat_to_rot(qw, qx, qy, qz).reshape(-1), [tx, ty, tz]]))
            valid += 1
    if not poses:
        raise ValueError(f"no poses in {path}")
    return np.asarray(ts), np.asarray(poses)


def rot_to_quat(R):
    """Rotation matrix -> (w, x, y, z), w >= 0 branch-stable (Shepperd)."""
    R = np.asarray(R).reshape(3, 3)
    tr = np.trace(R)
    if tr > 0:
        s = np.sqrt(tr + 1.0) * 2
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = np.zeros(4)
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    return q / np.linalg.norm(q)


def write_poses_tum(path, timestamps, poses):
    poses = np.asarray(poses).reshape(-1, 12)
    with open(path, "w") as f:
        for t, p in zip(timestamps, poses):
            w, x, y, z = rot_to_quat(p[:9])
            f.write(f"{t:.9f} {p[9]:.9f} {p[10]:.9f} {p[11]:.9f} {x:.9f} {y:.9f} {z:.9f} {w:.9f}\n")


# ---------------------------------------------------------------------------------------------------------------- PCD
_NP = {("F", 4): np.float32, ("F", 8): np.float64, ("U", 1): np.uint8, ("U", 2): np.uint16, ("U", 4): np.uint32,
       ("I", 1): np.int8, ("I", 2): np.int16, ("I", 4): np.int32}


def lzf_decompress(data, out_len):
    """liblzf stream: ctrl < 32 -> ctrl+1 literals; else back reference of length (ctrl >> 5) + 2 (7 -> + next byte)."""
    out = bytearray(out_len)
    i, o, n = 0, 0, len(data)
    while i < n:
        ctrl = data[i]; i += 1
        if ctrl < 32:
            ln = ctrl + 1
            out[o:o + ln] = data[i:i + ln]
            i += ln; o += ln
        else:
            ln = ctrl >> 5
            if ln == 7:
                ln += data[i]; i += 1
            ref = o - ((ctrl & 0x1f) << 8) - data[i] - 1
            i += 1
            for _ in range(ln + 2):            # may overlap: byte by byte
                out[o] = out[ref]; o += 1; ref += 1
    if o != out_len:
        raise ValueError("corrupt LZF stream")
    return bytes(out)


def lzf_compress_literal(data):
    """A valid (if uncompressing) LZF stream: literal runs only.  For writing test files / binary_compressed output."""
    out = bytearray()
    for i in range(0, len(data), 32):
        chunk = data[i:i + 32]
        out.append(len(chunk) - 1)
        out += chunk
    return bytes(out)


def load_pcd(path, fields=("x", "y", "z", "intensity")):
    """Returns float32 [n, len(found fields)] in the order of `fields` (missing ones dropped) and the list of names."""
    with open(path, "rb") as f:
        raw = f.read()
    hdr, pos = {}, 0
    while True:
        end = raw.index(b"\n", pos)
        line = raw[pos:end].decode("ascii", "replace").strip()
        pos = end + 1
        if not line or line.startswith("#"):
            continue
        key, _, val = line.partition(" ")
        hdr[key.upper()] = val.split()
        if key.upper() == "DATA":
            break
    names = hdr["FIELDS"]
    sizes = [int(v) for v in hdr["SIZE"]]
    types = hdr["TYPE"]
    counts = [int(v) for v in hdr.get("COUNT", ["1"] * len(names))]
    npts = int(hdr["POINTS"][0]) if "POINTS" in hdr else int(hdr["WIDTH"][0]) * int(hdr["HEIGHT"][0])
    mode = hdr["DATA"][0].lower()
    dt = np.dtype([(n if c == 1 else f"{n}", _NP[(t, s)], (c,) if c > 1 else ()) for n, s, t, c in zip(names, sizes, types, counts)])
    body = raw[pos:]
    if mode == "ascii":
        arr = np.loadtxt(body.decode("ascii").splitlines(), dtype=np.float64, ndmin=2)
        cols, k = {}, 0
        for n, c in zip(names, counts):
            cols[n] = arr[:, k]; k += c
    else:
        if mode == "binary_compressed":
            csz, usz = np.frombuffer(body[:8], np.uint32)
            blob = lzf_decompress(body[8:8 + int(csz)], int(usz))
            cols, k = {}, 0                       # stored field by field
            for n, s, t, c in zip(names, sizes, types, counts):
                a = np.frombuffer(blob, _NP[(t, s)], npts * c, k)
                cols[n] = a.reshape(npts, c)[:, 0] if c > 1 else a
                k += npts * c * s
        elif mode == "binary":
            rec = np.frombuffer(body, dt, npts)
            cols = {n: (rec[n][:, 0] if c > 1 else rec[n]) for n, c in zip(names, counts)}
        else:
            raise ValueError(f"unsupported PCD DATA mode {mode}")
    found = [n for n in fields if n in cols]
    return np.stack([np.asarray(cols[n], np.float32) for n in found], 1) if npts else np.zeros((0, len(found)), np.float32), found


def save_pcd(path, pts, fields=("x", "y", "z", "intensity"), mode="binary"):
    """float32 [n, len(fields)] -> PCD v0.7 (pcl::io::savePCDFileBinary's layout for `binary`)."""
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, len(fields))
    n = len(pts)
    hdr = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\n" + f"FIELDS {' '.join(fields)}\n" +
           f"SIZE {' '.join(['4'] * len(fields))}\nTYPE {' '.join(['F'] * len(fields))}\n" +
           f"COUNT {' '.join(['1'] * len(fields))}\nWIDTH {n}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {n}\nDATA {mode}\n")
    with open(path, "wb") as f:
        f.write(hdr.encode("ascii"))
        if mode == "ascii":
            for row in pts:
                f.write((" ".join(repr(float(v)) for v in row) + "\n").encode("ascii"))
        elif mode == "binary":
            f.write(pts.tobytes())
        elif mode == "binary_compressed":
            blob = np.ascontiguousarray(pts.T).tobytes()
            comp = lzf_compress_literal(blob)
            f.write(np.array([len(comp), len(blob)], np.uint32).tobytes() + comp)
        else:
            raise ValueError(mode)


def load_body_points(pcd_dir):
    """all_pcd_body/*.pcd sorted by the time stamp in the file name.  Returns (timestamps, [clouds [n_i, 4] x y z intensity])."""
    items = []
    for name in os.listdir(pcd_dir):
        if not name.endswith(".pcd"):
            continue
        ts = parse_timestamp_from_name(name)
        if ts is None:
            continue
        items.append((ts, os.path.join(pcd_dir, name)))
    if not items:
        raise ValueError(f"no pcd files in {pcd_dir}")
    items.sort(key=lambda kv: kv[0])
    return np.array([t for t, _ in items]), [load_pcd(p)[0] for _, p in items]


def load_dataset(dataset_path):
    """dataset_path/all_pcd_body/{lidar_poses.txt,*.pcd} -> dict(timestamps, poses [n,12], clouds) (handleLidarPoses +
    handleBodyPoints, src/dataset_io.cpp:186-283: pose m pairs with the m-th cloud in time order)."""
    _, poses = load_poses_tum(os.path.join(dataset_path, "all_pcd_body", "lidar_poses.txt"), 1)
    ts, clouds = load_body_points(os.path.join(dataset_path, "all_pcd_body"))
    n = min(len(poses), len(clouds))
    return dict(timestamps=ts[:n], poses=poses[:n], clouds=clouds[:n])


# ------------------------------------------------------------------------------------------------------------- COLMAP text
def write_images_txt(path, q_cw, t_cw):
    """q_cw [M,4] (w,x,y,z), t_cw [M,3] -> COLMAP images.txt body as the reference writes it (ids = row index)."""
    with open(path, "w") as f:
        for k, (q, t) in enumerate(zip(np.asarray(q_cw).reshape(-1, 4), np.asarray(t_cw).reshape(-1, 3))):
            f.write(f"{k} {q[0]:.6f} {q[1]:.6f} {q[2]:.6f} {q[3]:.6f} {t[0]:.6f} {t[1]:.6f} {t[2]:.6f} 1 {k}.jpg\n")
            f.write("0.0 0.0 -1\n")


def write_points3d_txt(path, xyz, rgb):
    with open(path, "w") as f:
        for i, (p, c) in enumerate(zip(np.asarray(xyz).reshape(-1, 3), np.asarray(rgb).reshape(-1, 3))):
            f.write(f"{i} {p[0]:.6f} {p[1]:.6f} {p[2]:.6f} {int(c[0])} {int(c[1])} {int(c[2])} 0\n")


# ------------------------------------------------------------------------------------------------------------- COLMAP database
COLMAP_MAX_NUM_IMAGES = (1 << 31) - 1


def image_ids_to_pair_id(id1, id2):
    """src/lvba_system.cpp:512-519 (COLMAP's own formula): ids ordered, id_small * (2^31 - 1) + id_large."""
    if id1 > id2:
        id1, id2 = id2, id1
    return int(id1) * COLMAP_MAX_NUM_IMAGES + int(id2)


def load_colmap_db(path, image_names, pairs):
    """LvbaSystem::loadFromColmapDB (src/lvba_system.cpp:510-685) without OpenCV/SiftGPU types.
    image_names: file names in the caller's image order (matched against images.name, as name2id upstream);
    pairs: [(i, j)] index pairs into that order.  Returns (keypoints, matches): keypoints[i] = float32 [n_i, cols] (x, y,
    then sigma / extremum if present) or an empty array when the image or its blob is missing; matches[k] = int32 [m, 2]
    inlier matches of two_view_geometries for pairs[k], columns in the order (i, j) of the pair (swapped back when the
    database stored the pair the other way round), out-of-range indices dropped."""
    import sqlite3
    con = sqlite3.connect(path)
    try:
        name2id = {name: int(iid) for iid, name in con.execute("SELECT image_id, name FROM images")}
        ids = [name2id.get(n, -1) for n in image_names]
        kps = []
        for iid in ids:
            row = con.execute("SELECT rows, cols, data FROM keypoints WHERE image_id=?", (iid,)).fetchone() if iid >= 0 else None
            if row is None or row[2] is None or len(row[2]) != row[0] * row[1] * 4:
                kps.append(np.zeros((0, 4), np.float32))
            else:
                kps.append(np.frombuffer(row[2], np.float32).reshape(row[0], row[1]).copy())
        out = []
        for i, j in pairs:
            m = np.zeros((0, 2), np.int32)
            a, b = ids[i], ids[j]
            if a >= 0 and b >= 0 and len(kps[i]) and len(kps[j]):
                row = con.execute("SELECT rows, cols, data FROM two_view_geometries WHERE pair_id=?",
                                  (image_ids_to_pair_id(a, b),)).fetchone()
                if row is not None and row[1] == 2 and row[2] is not None and row[0] > 0 and len(row[2]) == row[0] * 8:
                    m = np.frombuffer(row[2], np.uint32).reshape(-1, 2).astype(np.int64)
                    if a > b:                                   # stored as (smaller id, larger id)
                        m = m[:, ::-1]
                    ok = (m[:, 0] >= 0) & (m[:, 0] < len(kps[i])) & (m[:, 1] >= 0) & (m[:, 1] < len(kps[j]))
                    m = m[ok].astype(np.int32)
            out.append(m)
        return kps, out
    finally:
        con.close()
