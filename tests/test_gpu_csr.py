"""The sparse (CSR-by-post) graph form: same results as the oracle's dense masked matrix, bit for bit --
inputs (electrical + chemical), STDP through the row and the transpose index, spike-train presynaptic
cells, sharded handles.  Includes BASELINE configs[4]'s structure (4 lattices with radius-2 neighbourhoods,
one Poisson lattice per neuron lattice wired one-to-one, ring k -> k+1) at test size."""
import numpy as np
import pytest

import oracle_binding as ob
import parity

pytestmark = pytest.mark.gpu


def random_sparse_net(chemical, density=0.03, seed=5):
    lay = parity.Layout([(0, 12, 13), (4, 20, 20)], [(2, 6, 7)])
    net = parity.make_oracle(lay, st_kind=ob.ST_POISSON, chemical=chemical)
    nn, nc = net.n_neurons, net.n_cells
    rng = np.random.default_rng(seed)
    net["current_voltage"] = ob.uniform_array(seed, nn, -65.0, 30.0)
    net["gap_conductance"] = 10.0
    net["nt_flags"][:, 0] = 1
    net["nt_flags"][::3, 2] = 1
    net["rc_flags"][:, 0] = 1
    net["rc_flags"][::2, 2] = 1
    net["rc_g"][:, 0] = 3.0
    net["st_nt_flags"][:, 0] = 1
    net["st_chance_of_firing"] = 0.05
    net["st_seed"] = np.arange(7, 7 + nc, dtype=np.uint32)
    net.fill_graph(seed + 1, 0.5, 2.5, with_diagonal=True)
    net["connections"][...] = rng.random(net["connections"].shape) < density
    net["connections"][:, 5] = 0                     # a neuron without any input
    net["connections"][7, :] = 0                     # a neuron without any output
    net["weights"][...] *= net["connections"]
    net["do_plasticity"] = 1
    return net


def check(dn, net):
    rng = net.layout.ranges()
    for i, _, _ in net.layout.lattices:
        first, count, _ = rng[i]
        assert np.array_equal(dn.spike_history(i), net.spike_history[:, first:first + count])
        assert np.array_equal(parity.bits(dn.voltage_history(i)), parity.bits(net.voltage_history[:, first:first + count]))
    parity.assert_state_equal(net, parity.pull_state(dn, net))
    parity.assert_graph_equal(net, dn)


@pytest.mark.parametrize("chemical", [False, True])
def test_csr_handle_equals_oracle(snn, chemical):
    net = random_sparse_net(chemical)
    dn = parity.device_from_oracle(snn, net, csr=True)
    dn.set_history(voltage=True, spikes=True)
    dn.run(400)
    dn.run(200)
    net.run(600, voltage_history=True, spike_history=True)
    assert net.spike_history.sum() > 20
    check(dn, net)
    # 8 B per stored synapse + the 60 B of Izhikevich state per owned row that the one-launch step (k_step_csr) also moves, + 44 B
    # per row and live transmitter type with chemical synapses (here AMPA and GABA)
    # (+ 28 B per spike-train cell when the cells advance in that launch: electrical-only handles without weight updates)
    assert dn.input_kernel_bytes() == 8 * int(net["connections"].sum()) + (60 + (88 if chemical else 0)) * net.n_neurons
    dn.set_option("fused_step", 0)
    assert dn.input_kernel_bytes() == 8 * int(net["connections"].sum())         # k_inputs_csr alone
    dn.close()


def test_csr_and_dense_handles_agree(snn):
    net = random_sparse_net(True, density=0.2, seed=9)
    a = parity.device_from_oracle(snn, net, csr=True)
    b = parity.device_from_oracle(snn, net, csr=False)
    for h in (a, b):
        h.set_history(voltage=True, spikes=False)
        h.run(300)
    for i, _, _ in net.layout.lattices:
        assert np.array_equal(parity.bits(a.voltage_history(i)), parity.bits(b.voltage_history(i)))
    wa = parity.dense_from_csr(net, 0, net.n_neurons, a.get_graph_csr())
    wb, _ = b.get_graph_rows(0, net.n_tot)
    assert np.array_equal(parity.bits(wa), parity.bits(wb))
    a.close()
    b.close()


def c5_structure(side, seed=11):
    """4 neuron lattices (ids 0-3) side x side, internal radius-<=2 neighbourhood (<= 12 in-edges, w = 1),
    Poisson lattices ids 4-7 wired one-to-one (w = 1), lattice k -> k+1 (mod 4) one-to-one (w = 1)."""
    lay = parity.Layout([(k, side, side) for k in range(4)], [(4 + k, side, side) for k in range(4)])
    net = parity.make_oracle(lay, st_kind=ob.ST_POISSON)
    m = side * side
    nn = 4 * m
    net["current_voltage"] = ob.uniform_array(seed, nn, -65.0, 30.0)
    net["gap_conductance"] = 10.0
    net["st_chance_of_firing"] = 0.01
    net["st_seed"] = np.arange(1, 4 * m + 1, dtype=np.uint32)         # seeds = cell index + 1
    conn, w = net["connections"], net["weights"]
    offs = [(dr, dc) for dr in range(-2, 3) for dc in range(-2, 3) if 0 < dr * dr + dc * dc <= 4]
    assert len(offs) == 12
    for k in range(4):
        base = k * m
        for r in range(side):
            for c in range(side):
                q = base + r * side + c
                for dr, dc in offs:
                    rr, cc = r + dr, c + dc
                    if 0 <= rr < side and 0 <= cc < side:
                        conn[base + rr * side + cc, q] = 1
                conn[nn + k * m + r * side + c, q] = 1                 # its Poisson cell
                conn[((k - 1) % 4) * m + r * side + c, q] = 1          # lattice k-1 -> k
    w[...] = conn
    return net


@pytest.mark.parametrize("cells_in_step", [1, 0])
def test_c5_structure_csr(snn, cells_in_step):
    """cells_in_step 1 (default): the Poisson cells advance inside the rows' launch, rows reading the view of the step
    while the cells write the next one; 0: the cells' own launch after the neuron update."""
    net = c5_structure(12)
    assert net["connections"].sum(axis=0).max() == 14
    dn = parity.device_from_oracle(snn, net, csr=True)
    dn.set_option("cells_in_step", cells_in_step)
    dn.run(3)
    net.run(3)
    fired = np.where(np.arange(144) % 7 == 0, 1, -1).astype(np.int32)
    dn.set_attr(5, "last_firing_time", fired)        # cell state changed behind the stepper's back: the view is rebuilt
    net["st_last_firing_time"][144:288] = fired
    dn.set_history(voltage=True, spikes=True)
    dn.run(497)
    net.run(497, voltage_history=True, spike_history=True)
    assert net.spike_history.sum() > 50
    check(dn, net)
    assert dn.input_kernel_bytes() == 8 * int(net["connections"].sum()) + 60 * net.n_neurons + 28 * cells_in_step * net.n_cells
    dn.close()


@pytest.mark.parametrize("n_shards,halo,by_lattice", [(2, False, False), (4, False, False), (2, True, False), (4, True, False),
                                                      (8, True, False), (2, True, True), (3, True, True), (4, False, True)])
def test_c5_structure_csr_sharded(snn, n_shards, halo, by_lattice):
    """configs[4]'s multi-GPU shape on one device: sparse shard handles + the emulated exchange -- whole slots
    (all-gather) or, with a committed halo plan, exactly the neurons each peer's rows reference."""
    import torch
    from snn_amd import parallel
    side = 16 if by_lattice else 8               # slabs are multiples of 64 neurons: 16 x 16 lattices split 2 to 4 ways
    net = c5_structure(side)
    net["do_plasticity"] = 1
    handles = [parity.device_from_oracle(snn, net, shard=(r, n_shards), csr=True, by_lattice=by_lattice)
               for r in range(n_shards)]
    ex = parallel.LocalExchange(handles, torch.device("cuda", 0), halo=halo)
    # a shard that owns a slab of EVERY lattice always trades lists (whole ownerships until a halo plan is committed)
    assert all(p["mode"] == ("halo" if halo or by_lattice else "allgather") and p["planes"] == 1 for p in ex.plans)
    if by_lattice:
        owned = np.concatenate([h.owned for h in handles])
        assert np.array_equal(np.sort(owned), np.arange(net.n_neurons))          # a partition of the population
        assert all(len(h.ranges) in (0, 4) for h in handles)                     # one slab per lattice (or nothing: 3 shards)
        if halo:    # the ring edge k -> k+1 stays inside the shard: only the two lattice rows either side of a slab travel
            assert max(ex.bytes_per_step()) <= 4 * (4 * 4 * side + 4 * 4 * side // 32 + 8)
    read = []                                    # per handle: own neurons + what its rows read
    for r, h in enumerate(handles):
        k = np.zeros(net.n_neurons, bool)
        k[h.owned] = True
        if halo:
            for p in range(n_shards):
                if p != r:
                    k[h.halo_needs(p)] = True
            # radius-2 neighbourhoods + one ring edge per neuron: a strict subset of the other shards
            assert k.sum() < net.n_neurons
        else:
            k[:] = True
        read.append(k)
    for _ in range(300):
        ex.step()
    net.run(300, spike_history=True)
    assert net.spike_history.sum() > 20
    for h, k in zip(handles, read):
        st = parity.pull_state(h, net)
        for name in ("current_voltage", "is_spiking", "last_firing_time"):
            assert np.array_equal(parity.bits(st[name][k]), parity.bits(net[name][k])), name
        cells = h.cells_read()                   # a sparse shard advances the cells its own rows read
        assert cells.size == h.owned.size and cells.size < net.n_cells       # one Poisson cell per own neuron
        for name in ("st_last_firing_time", "st_seed"):
            assert np.array_equal(parity.bits(st[name][cells]), parity.bits(net[name][cells])), name
        o = h.owned
        assert np.array_equal(parity.bits(st["w_value"][o]), parity.bits(net["w_value"][o]))
        parity.assert_graph_equal(net, h)
        h.close()


def test_csr_validation(snn):
    dn = snn.DeviceNetwork()
    dn.add_lattice(0, 2, 2)
    dn.finalize(csr=True)
    with pytest.raises(snn.SnnError) as e:           # unsorted presynaptic indices
        dn.set_graph_csr([0, 2, 2, 2, 2], [3, 1], [1.0, 1.0])
    assert e.value.code == 11
    with pytest.raises(snn.SnnError) as e:           # index out of range
        dn.set_graph_csr([0, 1, 1, 1, 1], [9], [1.0])
    assert e.value.code == 10
    with pytest.raises(snn.SnnError) as e:           # dense API on a CSR handle
        dn.fill_graph_synthetic(1, 0.0, 1.0)
    assert e.value.code == 12
    dn.run(5)                                        # no graph set: no edges, still steps
    assert dn.clock == 5
    dn.close()


def test_range_set_shards_leave_foreign_rows_alone_and_keep_histories(snn):
    """Sparse shards BY LATTICE of lattices whose first index is not a multiple of 64: a shard's 64-row blocks then hold rows
    of neurons it does not own.  With chemical synapses only (no voltage on the wire) a foreign row stepped by mistake runs
    on its own and stamps last_firing_time; histories of a shard are indexed by GLOBAL 64-blocks."""
    import torch
    from snn_amd import parallel
    lay = parity.Layout([(0, 2, 6), (3, 8, 11), (5, 2, 4)], [(9, 3, 7)])
    net = parity.make_oracle(lay, st_kind=ob.ST_POISSON, chemical=True, electrical=False)
    nn, nc = net.n_neurons, net.n_cells
    rng = np.random.default_rng(77)
    net["current_voltage"] = ob.uniform_array(5, nn, -65.0, 29.0)
    net["nt_flags"][:, 0] = 1
    net["rc_flags"][:, 0] = 1
    net["rc_g"][:, 0] = 4.0
    net["st_nt_flags"][:, 0] = 1
    net["st_chance_of_firing"] = 0.1
    net["st_seed"] = np.arange(11, 11 + nc, dtype=np.uint32)
    net.fill_graph(3, 0.5, 2.5, with_diagonal=True)
    net["connections"][...] = rng.random(net["connections"].shape) < 0.08
    net["weights"][...] *= net["connections"]
    handles = [parity.device_from_oracle(snn, net, shard=(r, 3), csr=True, by_lattice=True) for r in range(3)]
    for h in handles:
        h.set_history(voltage=True, spikes=True)
    ex = parallel.LocalExchange(handles, torch.device("cuda", 0), halo=True)
    assert all(p["plane_id"] == [2] for p in ex.plans)           # one transmitter plane, no voltage
    for _ in range(250):
        ex.step()
    net.run(250, voltage_history=True, spike_history=True)
    assert net.spike_history.sum() > 20
    rngs = net.layout.ranges()
    for h in handles:
        st = parity.pull_state(h, net)
        parity.assert_shard_view_equal(h, st, net)
        own = np.zeros(nn, bool)
        own[h.owned] = True
        for i, _, _ in net.layout.lattices:
            first, count, _ = rngs[i]
            m = own[first:first + count]
            if not m.any():
                continue
            assert np.array_equal(h.spike_history(i)[:, m], net.spike_history[:, first:first + count][:, m])
            assert np.array_equal(parity.bits(h.voltage_history(i)[:, m]), parity.bits(net.voltage_history[:, first:first + count][:, m]))
        h.close()
