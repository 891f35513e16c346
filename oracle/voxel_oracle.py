"""CPU oracle of the voxel front-end that BUILDS the LiDAR-BA problem.  TEST INFRASTRUCTURE ONLY.

Literal restatement (paths relative to /root/reference) of
  cut_voxel                          include/BALM/bavoxel.hpp:799-836  (hash points into root voxels)
  OCTO_TREE_NODE::recut / cut_func   include/BALM/bavoxel.hpp:391-464  (adaptive octree, <= layer_limit = 2 splits)
  OCTO_TREE_NODE::judge_eigen        include/BALM/bavoxel.hpp:335-352  (planarity test lambda0/lambda2 vs eigen_ratio_array)
  OCTO_TREE_NODE::tras_opt           include/BALM/bavoxel.hpp:466-474  -> VOX_HESS::push_voxel :45-54
  OCTO_TREE_NODE::findCorrespondPoint include/BALM/bavoxel.hpp:320-333 (landmark -> plane lookup, src/lvba_system.cpp:1531-1565)
including its fp32 quirks: voxel keys from a FLOAT quotient with "-1 if negative" then C
truncation; voxel centres and quarter lengths stored as float; octant test `double > float`.
PINNED against the reference's own bavoxel.hpp / tools.hpp compiled with the Eigen / PCL stand-ins of oracle/shim
(oracle/_ref/libbalm_ref.so): tests/test_ref_pin.py finds the same roots, the same plane nodes at the same octant paths and
bit-identical per-frame clusters, and the same plane for every looked-up landmark; tests/golden/ref_voxel.npz carries the
reference's answers to the GPU box.  Also pinned by the hand-built cases of tests/test_voxel_oracle.py.
Traversal order of the reference's unordered_map is unspecified; voxels are reported sorted by (root key, path).
"""
from __future__ import annotations

import numpy as np

MIN_PS = 15          # bavoxel.hpp:24
LAYER_LIMIT = 2      # bavoxel.hpp:13
DEFAULT_EIGEN_RATIO = np.array([0.3, 0.1, 0.06, 0.03], dtype=np.float32)   # bavoxel.hpp:17

f32 = np.float32


def root_key(pw, voxel_size):
    """bavoxel.hpp:809-815."""
    key = []
    for j in range(3):
        loc = f32(pw[j] / voxel_size)
        if loc < 0:
            loc = f32(np.float64(loc) - 1.0)
        key.append(int(np.trunc(loc)))
    return tuple(key)


class Node:
    def __init__(self, win_size, layer, center, quater):
        self.layer = layer
        self.center = np.asarray(center, dtype=f32)       # float voxel_center[3]
        self.quater = f32(quater)                          # float quater_length
        self.pts = [[] for _ in range(win_size)]           # vec_orig
        self.sig = np.zeros((win_size, 10))                # sig_orig: Pxx Pxy Pxz Pyy Pyz Pzz vx vy vz N
        self.leaves = [None] * 8
        self.state = "UNKNOWN"
        self.plane_center = None
        self.plane_normal = None

    def push(self, f, p):
        """PointCluster::push in cloud order (tools.hpp:428-433)."""
        self.pts[f].append(p)
        s = self.sig[f]
        s[0] += p[0] * p[0]; s[1] += p[0] * p[1]; s[2] += p[0] * p[2]
        s[3] += p[1] * p[1]; s[4] += p[1] * p[2]; s[5] += p[2] * p[2]
        s[6] += p[0]; s[7] += p[1]; s[8] += p[2]; s[9] += 1


def _merged_cov(sig, poses):
    """judge_eigen's covMat: world-frame sum over non-empty frames, in frame order (bavoxel.hpp:337-344)."""
    from .balm_oracle import cluster_transform, unpack_clusters, unpack_poses
    Rs, ps = unpack_poses(poses)
    P, v, n = unpack_clusters(sig)
    sP, sv, sn = np.zeros((3, 3)), np.zeros(3), 0.0
    for i in range(sig.shape[0]):
        if n[i] > 0:
            P2, v2, n2 = cluster_transform(P[i], v[i], n[i], Rs[i], ps[i])
            sP += P2; sv += v2; sn += n2
    c = sv / sn
    return sP / sn - np.outer(c, c), c


def recut(node, poses, eigen_ratio, out_nodes):
    """bavoxel.hpp:391-464; appends every visited node to out_nodes."""
    out_nodes.append(node)
    win = node.sig.shape[0]
    if node.state == "UNKNOWN":
        if node.sig[:, 9].sum() < MIN_PS:                                  # :399-405
            node.state = "MID_NODE"
            return
        cov, c = _merged_cov(node.sig, poses)
        lam, U = np.linalg.eigh(cov)
        node.plane_center, node.plane_normal = c, U[:, 0]
        if not (lam[0] / lam[2] > float(eigen_ratio[node.layer])):         # judge_eigen :351-352
            node.state = "PLANE"
            return
        if node.layer == LAYER_LIMIT:                                      # :421-427
            node.state = "MID_NODE"
            return
        Rs = np.asarray(poses).reshape(-1, 12)
        for f in range(win):                                               # cut_func :357-389
            R, t = Rs[f, :9].reshape(3, 3), Rs[f, 9:]
            for p in node.pts[f]:
                pw = R @ p + t
                b = [1 if pw[k] > np.float64(node.center[k]) else 0 for k in range(3)]
                leaf = 4 * b[0] + 2 * b[1] + b[2]
                if node.leaves[leaf] is None:
                    cc = [f32(node.center[k] + f32(2 * b[k] - 1) * node.quater) for k in range(3)]
                    node.leaves[leaf] = Node(win, node.layer + 1, cc, f32(np.float64(node.quater) / 2.0))
                node.leaves[leaf].push(f, p)
        node.state = "SPLIT"
    for lf in node.leaves:
        if lf is not None:
            recut(lf, poses, eigen_ratio, out_nodes)


def build(frames, poses, voxel_size, eigen_ratio=DEFAULT_EIGEN_RATIO):
    """frames: list of [n_i, 3] float32 point arrays (body frame); poses [N, 12].
    Returns (surf_map {key: root Node}, voxels) where voxels is the list VOX_HESS::plvec_voxels would hold
    (each a [win_size, 10] sig_orig of an admitted PLANE node), sorted by (root key, octant path)."""
    win = len(frames)
    poses = np.asarray(poses, dtype=np.float64).reshape(-1, 12)
    surf_map = {}
    for f, pts in enumerate(frames):                                        # cut_voxel :799-836
        R, t = poses[f, :9].reshape(3, 3), poses[f, 9:]
        for p32 in np.asarray(pts, dtype=f32).reshape(-1, 3):
            p = p32.astype(np.float64)
            key = root_key(R @ p + t, voxel_size)
            if key not in surf_map:
                center = [f32((0.5 + key[j]) * voxel_size) for j in range(3)]
                surf_map[key] = Node(win, 0, center, f32(voxel_size / 4.0))
            surf_map[key].push(f, p)
    voxels = []
    for key in sorted(surf_map):
        nodes = []
        recut(surf_map[key], poses, eigen_ratio, nodes)

        def walk(n, path):                                                  # tras_opt :466-474 in octant order
            if n.state == "PLANE":
                if np.count_nonzero(n.sig[:, 9] != 0) >= 2:                 # push_voxel :45-54
                    voxels.append((key, path, n))
            else:
                for o, lf in enumerate(n.leaves):
                    if lf is not None:
                        walk(lf, path + (o,))
        walk(surf_map[key], ())
    return surf_map, voxels


def pack(voxels):
    """CSR arrays of lvba_balm_create from the admitted voxels."""
    off, idx, clu = [0], [], []
    for _, _, n in voxels:
        nz = np.nonzero(n.sig[:, 9] != 0)[0]
        idx.append(nz.astype(np.int32))
        clu.append(n.sig[nz])
        off.append(off[-1] + len(nz))
    return (np.asarray(off, np.int64), np.concatenate(idx) if idx else np.zeros(0, np.int32),
            np.concatenate(clu) if clu else np.zeros((0, 10)))


def find_plane(surf_map, X, voxel_size):
    """Landmark -> plane association of src/lvba_system.cpp:1531-1565 + findCorrespondPoint (bavoxel.hpp:320-333).
    Returns (n, d) or None."""
    if not np.all(np.isfinite(X)):
        return None
    key = []
    for j in range(3):                                                      # :1539-1544 (float arithmetic)
        loc = f32(X[j] / voxel_size)
        if loc < 0:
            loc = f32(loc - f32(1.0))
        key.append(int(np.trunc(loc)))
    node = surf_map.get(tuple(key))
    if node is None:
        return None
    while not (node.state == "PLANE" or node.layer >= LAYER_LIMIT):
        b = [1 if X[k] > np.float64(node.center[k]) else 0 for k in range(3)]
        nxt = node.leaves[4 * b[0] + 2 * b[1] + b[2]]
        if nxt is None:
            break
        node = nxt
    if node.state != "PLANE":
        return None
    n = node.plane_normal
    if not np.all(np.isfinite(n)) or np.linalg.norm(n) < 1e-6:
        return None
    n = n / np.linalg.norm(n)
    return n, -float(n @ node.plane_center)
