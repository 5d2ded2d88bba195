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
