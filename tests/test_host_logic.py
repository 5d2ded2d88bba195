"""Host-side logic that needs no GPU: shard geometry, exchange-buffer layout, synthetic data."""
import numpy as np

import snn_amd
from snn_amd import parallel


def test_shard_geometry_covers_population_in_wavefront_slots():
    for n, g in [(65536, 8), (65536, 1), (1000, 4), (81920, 8), (63, 2), (0, 2), (130, 3)]:
        stride, shards = parallel.shard_geometry(n, g)
        assert stride % 64 == 0 and stride * g >= n
        assert shards[0][0] == 0 and shards[-1][1] == n
        covered = 0
        for r, (b, e) in enumerate(shards):
            assert b == min(n, r * stride) and e - b <= stride and b <= e
            covered += e - b
        assert covered == n


def test_c2_geometry_is_exact_at_8_gpus():
    stride, shards = parallel.shard_geometry(256 * 256, 8)
    assert stride == 8192 and shards[7] == (57344, 65536)


def test_synthetic_uniform_is_counter_based():
    a = snn_amd.synthetic.uniform(2, 64, 0.5, 1.5, offset=1000)
    b = snn_amd.synthetic.uniform(2, 1064, 0.5, 1.5)[1000:]
    assert np.array_equal(a, b) and a.dtype == np.float32
    assert 0.5 <= a.min() and a.max() < 1.5
