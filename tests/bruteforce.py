"""Independent brute-force numpy definitions of the pointnet2 ops, used to pin the C oracle.

They deliberately share no code with oracle/mgar_oracle.c: distances are computed in
float64 with vectorised numpy, neighbours are found by sorting, FPS ties are resolved by an
explicit priority formula rather than by simulating the reference's reduction tree.
Exact comparisons are only made on inputs where float32 arithmetic is exact (small-integer
lattices); on random inputs pairs within a relative band of the radius are ignored.
"""
import numpy as np


def pair_d2(q, p):
    q = np.asarray(q, np.float64); p = np.asarray(p, np.float64)
    return ((q[:, None, :] - p[None, :, :]) ** 2).sum(-1)


def ball_query_rows(q, p, radius, nsample, strict=True, band=0.0):
    """Returns (rows, ambiguous): rows[i] = first nsample indices inside the ball (padded with
    the first), None if empty; ambiguous[i] = True if some pair sits within `band` of r^2."""
    d2 = pair_d2(q, p)
    r2 = float(np.float32(radius) * np.float32(radius))
    inside = d2 < r2 if strict else d2 <= r2
    amb = (np.abs(d2 - r2) <= band * r2).any(1)
    rows = []
    for i in range(q.shape[0]):
        hits = np.nonzero(inside[i])[0][:nsample]
        if hits.size == 0:
            rows.append(None)
        else:
            rows.append(np.concatenate([hits, np.full(nsample - hits.size, hits[0])]).astype(np.int32))
    return rows, amb


def bitrev(v, bits):
    r = 0
    for _ in range(bits):
        r = (r << 1) | (v & 1)
        v >>= 1
    return r


def fps_lattice(p, m, block_size):
    """Farthest point sampling on exact (integer-lattice) data with the reference's tie rule
    written as a closed form: among equal maxima the winner minimises
    (bitrev(k mod bs), k // bs)."""
    p = np.asarray(p, np.float64)
    n = p.shape[0]
    bits = int(np.log2(block_size))
    assert 1 << bits == block_size
    L = (n + block_size - 1) // block_size
    prio = np.array([bitrev(k % block_size, bits) * L + k // block_size for k in range(n)])
    temp = np.full(n, 1e10)
    out = [0]
    old = 0
    for _ in range(1, m):
        d = ((p - p[old]) ** 2).sum(1)
        temp = np.minimum(temp, d)
        best = temp.max()
        cand = np.nonzero(temp == best)[0]
        old = int(cand[np.argmin(prio[cand])])
        out.append(old)
    return np.array(out, np.int32), temp.astype(np.float32)


def three_nn_lattice(u, k):
    d2 = pair_d2(u, k)
    m = k.shape[0]
    idx = np.zeros((u.shape[0], 3), np.int32)
    dist = np.full((u.shape[0], 3), np.inf, np.float32)
    order = np.lexsort((np.broadcast_to(np.arange(m), d2.shape), d2), axis=1) if m else None
    for i in range(u.shape[0]):
        for j in range(min(3, m)):
            idx[i, j] = order[i, j]
            dist[i, j] = d2[i, order[i, j]]
    return dist, idx
