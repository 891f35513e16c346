"""The block cyclic reduction of csrc/bcr.hip as an algorithm, on the CPU (numpy): the one-launch-per-level form -- every even block
row's update computed from the level's INPUT coupling blocks (two L arrays, read one / write the other), both odd neighbours
inverted by the even row itself, T1 / T2 / t of an odd row kept for the back substitution, the levels from stride 1 up to the
last one that leaves row 0 alone, then the back substitution from the largest stride down -- against a dense solve.  What the GPU
tests hold the kernels to (tests/test_gpu_visual.py) is this scheme; here its index arithmetic is checked for block-row counts
that are not powers of two, including the ones with a missing right neighbour at several levels."""
import numpy as np
import pytest


def bcr_solve(D, L, rhs):
    """D [nb, b, b] diagonal blocks, L [nb, b, b] with L[r] = S[r, r - 1] (L[0] unused), rhs [nb, b].  Returns x [nb, b]."""
    nb, b = rhs.shape
    D = D.copy(); rhs = rhs.copy()
    Lbuf = [L.copy(), np.zeros_like(L)]
    cur = 0
    T1 = np.zeros_like(L); T2 = np.zeros_like(L); t = np.zeros_like(rhs)
    s, top = 1, 0
    while s < nb:
        Ls, Ld = Lbuf[cur], Lbuf[cur ^ 1]
        newD, newrhs = {}, {}
        for r in range(0, nb, 2 * s):                       # one "workgroup" per even row; reads only level inputs
            il, ir, q = r - s, r + s, r + 2 * s
            dD = np.zeros((b, b)); dL = np.zeros((b, b)); dr = np.zeros(b)
            if il >= 0:
                inv = np.linalg.inv(D[il])
                t1l, t2l, tl = inv @ Ls[il], inv @ Ls[r].T, inv @ rhs[il]
                dD += Ls[r] @ t2l
                dL = Ls[r] @ t1l
                dr += Ls[r] @ tl
            if ir < nb:
                inv = np.linalg.inv(D[ir])
                t1r = inv @ Ls[ir]
                t2r = inv @ Ls[q].T if q < nb else np.zeros((b, b))
                tr = inv @ rhs[ir]
                T1[ir], T2[ir], t[ir] = t1r, t2r, tr        # stored by the LEFT even neighbour
                dD += Ls[ir].T @ t1r
                dr += Ls[ir].T @ tr
            newD[r] = D[r] - dD
            newrhs[r] = rhs[r] - dr
            Ld[r] = -dL if (il >= 0 and r - 2 * s >= 0) else 0.0
        for r in newD:                                       # the even rows' own blocks: nobody else reads them at this level
            D[r], rhs[r] = newD[r], newrhs[r]
        cur ^= 1
        top = s
        s *= 2
    x = np.zeros_like(rhs)
    x[0] = np.linalg.solve(D[0], rhs[0])
    s = top
    while s >= 1:
        for i in range(s, nb, 2 * s):
            x[i] = t[i] - T1[i] @ x[i - s] - (T2[i] @ x[i + s] if i + s < nb else 0.0)
        s //= 2
    return x


@pytest.mark.parametrize("nb", [2, 3, 5, 8, 13, 16, 37, 100, 400])
def test_one_launch_levels_equal_dense_solve(nb):
    rng = np.random.default_rng(nb)
    b = 6
    L = 0.3 * rng.standard_normal((nb, b, b))
    L[0] = 0.0
    A = np.zeros((nb * b, nb * b))
    for r in range(nb):
        M = rng.standard_normal((b, b))
        A[r * b:(r + 1) * b, r * b:(r + 1) * b] = M @ M.T + 4.0 * np.eye(b)
        if r > 0:
            A[r * b:(r + 1) * b, (r - 1) * b:r * b] = L[r]
            A[(r - 1) * b:r * b, r * b:(r + 1) * b] = L[r].T
    assert np.linalg.eigvalsh(A).min() > 0.5
    D = np.stack([A[r * b:(r + 1) * b, r * b:(r + 1) * b] for r in range(nb)])
    rhs = rng.standard_normal((nb, b))
    x = bcr_solve(D, L, rhs)
    ref = np.linalg.solve(A, rhs.reshape(-1)).reshape(nb, b)
    assert np.abs(x - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max())
