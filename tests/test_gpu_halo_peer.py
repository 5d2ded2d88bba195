"""The PEER form of a library-driven run on sparse shard handles (include/snn_amd.h, snn_p2p_*): ONE launch per step and no
collective -- border rows store {voltage, tag | spike} granules straight into the peers' receive sets, the next step's rows
read them when the tag is the step's, done counters say when a set may be overwritten.  2 - 8 shard handles of one process on
one GPU (peer = same device), every rank a host thread inside snn_run_sharded; against the oracle, and against the same
run over the collective."""
import numpy as np
import pytest

import oracle_binding as ob
import parity
from test_gpu_csr import c5_structure
from test_gpu_library_loop_threads import collectives, run_ranks  # noqa: F401  (fixture)

pytestmark = [pytest.mark.gpu, pytest.mark.emulated_ranks]


def check_against_oracle(handles, net, steps):
    for h in handles:
        assert h.clock == steps
        st = parity.pull_state(h, net)
        parity.assert_shard_view_equal(h, st, net)
        cells = h.cells_read()
        for name in ("st_last_firing_time", "st_seed"):
            assert np.array_equal(parity.bits(st[name][cells]), parity.bits(net[name][cells])), name
        assert np.array_equal(parity.bits(st["w_value"][h.owned]), parity.bits(net["w_value"][h.owned]))


@pytest.mark.timeout(300)
@pytest.mark.parametrize("n_shards,by_lattice,side", [(2, True, 16), (4, True, 16), (3, False, 8), (8, False, 8), (8, True, 32)])
def test_peer_form_equals_the_oracle(snn, collectives, n_shards, by_lattice, side):
    from snn_amd import parallel
    net = c5_structure(side)
    handles = [parity.device_from_oracle(snn, net, shard=(r, n_shards), csr=True, by_lattice=by_lattice) for r in range(n_shards)]
    parallel.wire_halo_lists(handles)
    assert parallel.connect_peers(handles)
    for h in handles:
        # (8 ranks as threads of one process on ONE device wait for each other's launches through shared hardware queues: a
        # generous limit -- once in a full run of the suite the 1 << 20 of the smaller cases was not enough)
        h.set_option("halo_peer_spin_limit", 1 << (20 if n_shards <= 4 else 24))
    tc = collectives(n_shards)
    # (8 ranks on one device: their launches share hardware queues, where a kernel that waits for a neighbour's values may sit in
    # front of the kernel that produces them -- a limit of this emulation, not of one rank per GPU; those cases step call by call,
    # which keeps every rank's launches of one step ahead of anybody's next)
    calls = [150, 1, 49] if n_shards <= 4 else [1] * 60
    run_ranks(handles, tc, calls)
    steps = sum(calls)
    net.run(steps, spike_history=True)
    assert net.spike_history.sum() > 5 or steps < 100
    for h in handles:
        plan = h.exchange_plan()
        if h.owned.size and int(plan["send_words"]) + int(plan["recv_words"]):
            assert h.stat("halo_peer_steps") == steps and h.stat("steps_sparse_one_launch") == steps, \
                (h.stat("halo_peer_steps"), h.stat("steps_sparse_one_launch"), h.stat("steps_sparse_split"), steps)
    check_against_oracle(handles, net, steps)
    for h in handles:
        h.close()


def with_transmitters(net, seed, electrical=True):
    """chemical synapses on the C5 structure: two thirds of the neurons release a transmitter of their own choice of types, every
    neuron has receptors, a third of the Poisson cells release too (the halo then carries up to three transmitter planes)"""
    rng = np.random.default_rng(seed)
    nn, nc = net.n_neurons, net.n_cells
    net.electrical, net.chemical = electrical, True
    net["nt_flags"][...] = (rng.random((nn, 3)) < 0.5) & (rng.random((nn, 1)) < 0.67)
    net["nt_t"][...] = rng.random((nn, 3)).astype(np.float32) * net["nt_flags"]
    net["rc_flags"][...] = rng.random((nn, 3)) < 0.6
    net["st_nt_flags"][...] = (rng.random((nc, 3)) < 0.5) & (rng.random((nc, 1)) < 0.33)
    net["weights"][...] *= ob.uniform_array(seed + 1, net["weights"].size, 0.2, 1.2).reshape(net["weights"].shape)
    return net


@pytest.mark.timeout(600)
@pytest.mark.parametrize("n_shards,by_lattice,electrical,delay", [(2, True, True, 0), (4, True, True, 0), (3, False, False, 0), (4, False, True, 0),
                                                                  (2, True, True, 30), (4, True, True, 30), (3, False, True, 12),
                                                                  (8, True, True, 0), (8, False, True, 8)])
def test_peer_form_with_transmitter_planes_and_injected_delays(snn, collectives, n_shards, by_lattice, electrical, delay):
    """BASELINE configs[4]'s network with chemical synapses (the reference's network chemical inputs, neuron/gpu_lattices/mod.rs:
    1278-1382): a halo neuron travels as one granule per plane -- voltage and the transmitter types some neuron releases -- and a
    transmitter granule carries "this neuron releases the type" in its tag word.  delay > 0: every store of granules, every row's
    first poll and every done announcement waits a pseudo-random time of up to delay x 3 us first (halo_peer_delay), different
    per rank, wavefront and step -- what first contact with a real interconnect would do to the protocol."""
    from snn_amd import parallel
    net = with_transmitters(c5_structure(16), 100 + n_shards, electrical)
    handles = [parity.device_from_oracle(snn, net, shard=(r, n_shards), csr=True, by_lattice=by_lattice) for r in range(n_shards)]
    parallel.wire_halo_lists(handles)
    assert parallel.connect_peers(handles)
    for h in handles:
        h.set_option("halo_peer_spin_limit", 1 << 24)
        h.set_option("halo_peer_delay", delay)
    tc = collectives(n_shards)
    calls = [40, 1, 19] if n_shards <= 4 else [1] * 30
    run_ranks(handles, tc, calls)
    steps = sum(calls)
    net.n_threads = 8
    net.run(steps)
    busy = 0
    for h in handles:
        plan = h.exchange_plan()
        assert plan["planes"] >= (2 if electrical else 1)
        if h.owned.size and int(plan["send_words"]) + int(plan["recv_words"]):
            busy += 1
            assert h.stat("halo_peer_steps") == steps and h.stat("steps_sparse_one_launch") == steps, \
                (h.stat("halo_peer_steps"), h.stat("steps_sparse_one_launch"), h.stat("steps_sparse_split"), steps)
    assert busy >= 2
    check_against_oracle(handles, net, steps)
    for h in handles:
        st = parity.pull_state(h, net)
        for name in ("rc_r", "rc_current"):
            assert np.array_equal(parity.bits(st[name][h.owned]), parity.bits(net[name][h.owned])), name
        h.close()


@pytest.mark.timeout(300)
def test_injected_delays_leave_the_electrical_peer_form_bit_exact(snn, collectives):
    from snn_amd import parallel
    n_shards = 4
    net = c5_structure(16)
    handles = [parity.device_from_oracle(snn, net, shard=(r, n_shards), csr=True, by_lattice=True) for r in range(n_shards)]
    parallel.wire_halo_lists(handles)
    assert parallel.connect_peers(handles)
    for h in handles:
        h.set_option("halo_peer_spin_limit", 1 << 24)
        h.set_option("halo_peer_delay", 33)                 # up to 100 us per call site
    run_ranks(handles, collectives(n_shards), [80, 20])
    net.run(100)
    assert all(h.stat("halo_peer_steps") == 100 for h in handles)
    check_against_oracle(handles, net, 100)
    for h in handles:
        h.close()


@pytest.mark.timeout(120)
def test_peer_form_between_other_steps_and_switched_off(snn, collectives):
    """runs of the peer form alternate with host-driven steps (which move the halo through the ordinary segments) and with the
    collective form (option halo_peer 0): the mirror, the tags and the done counters carry over"""
    import torch
    from snn_amd import parallel
    n_shards = 4
    net = c5_structure(16)
    handles = [parity.device_from_oracle(snn, net, shard=(r, n_shards), csr=True, by_lattice=True) for r in range(n_shards)]
    ex = parallel.LocalExchange(handles, torch.device("cuda", 0), halo=True)       # (commits the halo lists: before connecting)
    assert parallel.connect_peers(handles)
    for h in handles:
        # (2^24 polls, as the other multi-phase tests: the four ranks are Python threads whose collectives are Python callbacks --
        # after the collective phase one of them can reach its first peer-form launch a second behind the others; with 2^20
        # the last phase gave up once in five fresh processes in round 6)
        h.set_option("halo_peer_spin_limit", 1 << 24)
    tc = collectives(n_shards)
    run_ranks(handles, tc, [60])
    for _ in range(7):                                   # host-driven steps in between
        for h in handles:
            h.step_begin_local()
            h.step_begin()
        ex.exchange()
        for h in handles:
            h.step_end()
    run_ranks(handles, tc, [40])
    for h in handles:
        h.set_option("halo_peer", 0)
    run_ranks(handles, tc, [30])
    for h in handles:
        h.set_option("halo_peer", 1)
    run_ranks(handles, tc, [25])
    net.n_threads = 8
    net.run(162)
    assert all(h.stat("halo_peer_steps") in (0, 125) for h in handles) and any(h.stat("halo_peer_steps") == 125 for h in handles)
    check_against_oracle(handles, net, 162)
    for h in handles:
        h.close()


def test_a_missing_peer_ends_the_run_with_an_error(snn, collectives):
    """a rank whose neighbour never steps gives up after the spin limit: SNN_ERR_WAIT, no hang"""
    from snn_amd import parallel
    net = c5_structure(8)
    handles = [parity.device_from_oracle(snn, net, shard=(r, 2), csr=True, by_lattice=False) for r in range(2)]
    parallel.wire_halo_lists(handles)
    parallel.connect_peers(handles)
    handles[0].set_option("halo_peer_spin_limit", 1 << 14)
    with pytest.raises(snn.SnnError) as e:
        handles[0].run_sharded_without_exchange(3)            # rank 1 never runs: its values for step 1 and its done counter never come
    assert e.value.code == 6                             # SNN_ERR_WAIT
    for h in handles:
        h.close()
