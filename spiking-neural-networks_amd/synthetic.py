"""Counter-based synthetic inputs (numpy twin of csrc/snn_math.hpp `uniform_from_hash`).

BASELINE.md's generator: value(seed, index) = lo + (hi - lo) * u24(splitmix64(seed, index)).
The device-side graph generator (`snn_fill_graph_synthetic`) uses the same function, so a
benchmark-size matrix is defined by (seed, lo, hi) alone and never crosses PCIe.
"""
import numpy as np


def hash32(seed, index):
    idx = np.asarray(index, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = idx + np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15)
        x = x ^ (x >> np.uint64(30))
        x = x * np.uint64(0xBF58476D1CE4E5B9)
        x = x ^ (x >> np.uint64(27))
        x = x * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    return (x >> np.uint64(32)).astype(np.uint32)


def uniform(seed, count, lo, hi, offset=0):
    """float32 array of `count` values for indices offset .. offset+count-1"""
    h = hash32(seed, np.arange(offset, offset + count, dtype=np.uint64))
    u = (h >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return (np.float32(lo) + (np.float32(hi) - np.float32(lo)) * u).astype(np.float32)


def c5_csr(side, q0=None, q1=None, posts=None, cols=None):
    """CSR rows (posts q0..q1, or the ascending global indices `posts`) of BASELINE configs[4]: four side x side
    (side x cols when `cols` is given: the weak-scaling form of bench.py) neuron
    lattices (ids 0-3), internal radius-<=2 neighbourhood (<= 12 in-edges, w = 1), one Poisson lattice per
    neuron lattice (ids 4-7) wired one-to-one (w = 1), lattice k -> k+1 (mod 4) one-to-one (w = 1).
    Returns (row_ptr uint64, pre_index uint32 ascending per row, weights float32)."""
    cols_ = side if cols is None else cols
    m = side * cols_
    nn = 4 * m
    q0 = 0 if q0 is None else q0
    q1 = nn if q1 is None else q1
    q = np.arange(q0, q1, dtype=np.int64) if posts is None else np.asarray(posts, dtype=np.int64)
    k, rem = q // m, q % m
    r, c = rem // cols_, rem % cols_
    offs = [(dr, dc) for dr in range(-2, 3) for dc in range(-2, 3) if 0 < dr * dr + dc * dc <= 4]
    sentinel = np.iinfo(np.int64).max
    entries = []
    for dr, dc in offs:
        rr, cc = r + dr, c + dc
        ok = (rr >= 0) & (rr < side) & (cc >= 0) & (cc < cols_)
        entries.append(np.where(ok, k * m + rr * cols_ + cc, sentinel))
    entries.append(((k - 1) % 4) * m + rem)            # previous lattice, same position
    entries.append(nn + k * m + rem)                   # its Poisson cell
    pre = np.sort(np.stack(entries, axis=1), axis=1)
    valid = pre != sentinel
    row_ptr = np.concatenate([[0], np.cumsum(valid.sum(axis=1))]).astype(np.uint64)
    pre_index = pre[valid].astype(np.uint32)
    return row_ptr, pre_index, np.ones(pre_index.size, np.float32)
