"""The step image of a sparse handle (csrc/snn_kernels_csr.hpp "STEP IMAGE", k_step_csr_img): 16-byte records of two entries and
the slices' presynaptic windows staged in LDS.  Bit for bit the oracle -- and the plain one-launch step -- on: the structure of
BASELINE configs[4] (every slice staged), unstructured graphs (no slice staged: plain codes through the records), graphs that mix
both, ragged and empty rows, rows far longer than the record ring, chunk crossings, Rate cells that have never fired (the view's
second word), and a handle whose weights change in between (plasticity on, then off: the records are rebuilt)."""
import numpy as np
import pytest

import oracle_binding as ob
import parity
from test_gpu_csr import c5_structure

pytestmark = pytest.mark.gpu


def run_both(snn, net, steps, expect_staged=None, histories=True):
    """the same network with and without the image, against the oracle; returns the image handle's statistics"""
    ref = None
    stats = {}
    for image in (1, 0):
        dn = parity.device_from_oracle(snn, net, csr=True)
        dn.set_option("csr_image", image)
        if histories:
            dn.set_history(voltage=True, spikes=True)
        dn.run(steps // 2)
        dn.run(steps - steps // 2)
        st = parity.pull_state(dn, net)
        hist = [(parity.bits(dn.voltage_history(i)), dn.spike_history(i)) for i, _, _ in net.layout.lattices] if histories else []
        if image:
            stats = {k: dn.stat(k) for k in ("steps_sparse_image", "steps_sparse_one_launch", "image_staged_slices")}
            ref = (st, hist)
        else:
            assert dn.stat("steps_sparse_image") == 0
            for k in ref[0]:
                assert np.array_equal(parity.bits(ref[0][k]), parity.bits(st[k])), k
            for (va, sa), (vb, sb) in zip(ref[1], hist):
                assert np.array_equal(va, vb) and np.array_equal(sa, sb)
        dn.close()
    onet = net
    onet.run(steps, voltage_history=histories, spike_history=histories)
    parity.assert_state_equal(onet, ref[0])
    rng = net.layout.ranges()
    for (i, _, _), (v, s) in zip(net.layout.lattices, ref[1]):
        first, count, _ = rng[i]
        assert np.array_equal(s, onet.spike_history[:, first:first + count])
        assert np.array_equal(v, parity.bits(onet.voltage_history[:, first:first + count]))
    assert stats["steps_sparse_image"] == steps == stats["steps_sparse_one_launch"]
    if expect_staged is not None:
        assert stats["image_staged_slices"] == expect_staged, stats
    return stats


def test_the_structure_of_configs4_is_staged_slice_by_slice(snn):
    net = c5_structure(24)                               # 4 x 576 neurons + as many Poisson cells; 14 entries per interior row
    st = run_both(snn, net, 60, expect_staged=(4 * 576 + 63) // 64)
    assert st["image_staged_slices"] == 36


def sparse(layout, st_layout, seed, st_kind=ob.ST_RATE, density=0.04, band=None, long_rows=()):
    """a graph drawn at random; band = neurons only read sources within that distance (a structured graph: stageable)"""
    net = parity.make_oracle(parity.Layout(layout, st_layout), st_kind=st_kind)
    nn, nc = net.n_neurons, net.n_cells
    rng = np.random.default_rng(seed)
    net["current_voltage"] = ob.uniform_array(seed, nn, -65.0, 30.0)
    net["gap_conductance"] = 8.0 + 4.0 * rng.random(nn).astype(np.float32)          # per-neuron conductances
    net.fill_graph(seed + 1, 0.2, 1.2, with_diagonal=True)
    conn = rng.random(net["connections"].shape) < density
    if band is not None:
        p, q = np.meshgrid(np.arange(net.n_tot), np.arange(nn), indexing="ij")
        conn &= (np.abs(p - q) <= band) | (p >= nn)
    for q in long_rows:
        conn[:, q] = rng.random(net.n_tot) < 0.6
    net["connections"][...] = conn
    net["weights"][...] *= net["connections"]
    if nc:
        if st_kind == ob.ST_RATE:
            net["st_rate"] = np.where(np.arange(nc) % 3 == 0, 0.0, 0.4 + 0.2 * (np.arange(nc) % 5)).astype(np.float32)   # a third never fires
        else:
            net["st_chance_of_firing"] = 0.03
            net["st_seed"] = np.arange(11, 11 + nc, dtype=np.uint32)
    return net


def test_an_unstructured_graph_goes_through_the_records_with_plain_codes(snn):
    net = sparse([(0, 30, 30), (3, 17, 19)], [(5, 6, 6)], seed=21, density=0.05)
    st = run_both(snn, net, 80)
    assert st["image_staged_slices"] == 0                                # (3 of 64 sources per piece: sixteen pieces never cover a slice)


def test_staged_and_unstaged_slices_in_one_launch_with_ragged_and_long_rows(snn):
    # banded rows (stageable) + a few rows that read 60 % of everything (their slices are not), an empty row, rows past a ragged end
    # (a window holds 1024 words: a network must be larger than that for a slice's sources not to fit)
    net = sparse([(0, 40, 40), (2, 9, 11)], [(4, 5, 7)], seed=33, density=0.5, band=6, long_rows=(3, 300, 1501))
    net["connections"][:, 77] = 0
    net["weights"][:, 77] = 0
    st = run_both(snn, net, 40)
    n_slices = (net.n_neurons + 63) // 64
    assert 0 < st["image_staged_slices"] < n_slices
    assert net["connections"].sum(axis=0).max() > 500                   # far past the ring of four records


@pytest.mark.parametrize("st_kind", [ob.ST_RATE, ob.ST_POISSON])
def test_cells_through_the_window_fired_and_never_fired(snn, st_kind):
    net = sparse([(0, 16, 16)], [(1, 8, 8)], seed=5, st_kind=st_kind, density=0.6, band=3)
    net["connections"][net.n_neurons:, :] = np.random.default_rng(1).random((net.n_cells, net.n_neurons)) < 0.1
    net["weights"][net.n_neurons:, :] = net["connections"][net.n_neurons:, :] * np.float32(1.5)
    run_both(snn, net, 90)


def test_weights_that_change_in_between_rebuild_the_records(snn):
    net = sparse([(0, 14, 14)], [(1, 3, 3)], seed=8, density=0.7, band=4)
    dn = parity.device_from_oracle(snn, net, csr=True)
    dn.run(20)
    assert dn.stat("steps_sparse_image") == 20
    dn.set_plasticity(0, do_plasticity=True)             # STDP on: the plain kernels step, weights move
    net["do_plasticity"] = 1
    dn.run(40)
    assert dn.stat("steps_sparse_image") == 20
    dn.set_plasticity(0, do_plasticity=False)            # off again: the image comes back with the NEW weights
    net["do_plasticity"] = 0
    dn.run(30)
    assert dn.stat("steps_sparse_image") == 50
    net.run(20)
    net["do_plasticity"] = 1
    net.run(40)
    net["do_plasticity"] = 0
    net.run(30)
    assert not np.array_equal(net["weights"], sparse([(0, 14, 14)], [(1, 3, 3)], seed=8, density=0.7, band=4)["weights"])
    parity.assert_state_equal(net, parity.pull_state(dn, net))
    parity.assert_graph_equal(net, dn)
    # a checkpoint restore puts old weights back: again a rebuild
    dn.checkpoint()
    dn.set_plasticity(0, do_plasticity=True)
    dn.run(10)
    dn.set_plasticity(0, do_plasticity=False)
    dn.restore_checkpoint()
    dn.run(15)
    net.run(15)
    parity.assert_state_equal(net, parity.pull_state(dn, net))
    dn.close()
