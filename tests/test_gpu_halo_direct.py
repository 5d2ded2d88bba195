"""Sparse shard handles in a library-driven run (snn_run_sharded / snn_run_sharded_custom): the rows gather the halo from the
received segments themselves, two sets of segments alternate, nothing is unpacked before the next step's rows and the step is
one or two launches (k_step_csr over the border slices, then over the interior slices with the cells, the mirror copy of
the previous step's arrivals and the bitmap clearing behind them).

G handles of ONE process, each driven by its own thread through snn_run_sharded_custom; the exchange function copies the
peers' CURRENT outgoing segments (snn_exchange_plan_get inside the function) device to device.  Against the oracle."""
import os
import threading

import numpy as np
import pytest

import parity
from test_gpu_csr import c5_structure

pytestmark = pytest.mark.gpu


def run_in_lockstep(handles, calls):
    """every handle runs `calls` (a list of step counts) with the library's loop; a failing thread breaks the barrier"""
    import torch
    from snn_amd import parallel
    g = len(handles)
    dev = torch.device("cuda", 0)
    barrier = threading.Barrier(g)
    static = [h.exchange_plan() for h in handles]          # counts and offsets (the pointers change from step to step)
    current = [None] * g
    errors = []

    def exchange_of(r):
        def exchange(_stream):
            torch.cuda.synchronize()                         # every handle's step so far (one device)
            current[r] = handles[r].exchange_plan()
            barrier.wait()
            mine = current[r]
            recv = parallel.device_words(mine["recv"], mine["recv_words"], dev)
            for p in range(g):
                n = int(static[p]["send_count"][r]) if p != r else 0
                if n == 0:
                    continue
                assert n == int(static[r]["recv_count"][p])
                send = parallel.device_words(current[p]["send"], current[p]["send_words"], dev)
                so, ro = int(static[p]["send_offset"][r]), int(static[r]["recv_offset"][p])
                recv[ro:ro + n].copy_(send[so:so + n])
            torch.cuda.synchronize()
            barrier.wait()                                   # nobody packs its next step while a peer still copies
        return exchange

    def work(r):
        try:
            for steps in calls:
                handles[r].run_sharded_custom(exchange_of(r), steps)
        except BaseException as e:       # noqa: BLE001 -- reported by the main thread
            errors.append(e)
            barrier.abort()

    threads = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(g)]
    for t in threads:
        t.start()
    try:
        for t in threads:
            t.join()
    except BaseException:                # (pytest-timeout interrupts the join: release the ranks, do not outlive the test)
        barrier.abort()
        raise
    if errors:
        raise errors[0]


@pytest.mark.timeout(600)
@pytest.mark.parametrize("n_shards,by_lattice,direct", [(2, True, 2), (4, True, 2), (3, True, 2), (2, False, 2), (4, False, 2),
                                                        (8, False, 2), (4, True, 0)])
def test_direct_halo_runs_equal_the_oracle(snn, n_shards, by_lattice, direct):
    import torch
    from snn_amd import parallel
    net = c5_structure(16 if by_lattice else 8)
    handles = [parity.device_from_oracle(snn, net, shard=(r, n_shards), csr=True, by_lattice=by_lattice) for r in range(n_shards)]
    for h in handles:
        h.set_option("halo_direct", direct)
    ex = parallel.LocalExchange(handles, torch.device("cuda", 0), halo=True)          # wires and commits the halo lists
    run_in_lockstep(handles, [40, 1, 59])
    for _ in range(7):                                      # host-driven steps in between: the ordinary protocol
        ex.step()
    run_in_lockstep(handles, [93])
    net.run(200, spike_history=True)
    assert net.spike_history.sum() > (20 if by_lattice else 5)
    for r, h in enumerate(handles):
        if h.owned.size:                                    # (a shard without neurons may or may not take the direct form)
            assert h.stat("halo_direct_steps") == (193 if direct else 0)
            # every step read the step image (round 6): in the direct runs the one whose halo sources are words of the receive
            # buffer, in the host-driven steps the plain one, the border launches packing as they go
            assert h.stat("steps_sparse_image") == 200 and h.stat("image_staged_slices_direct" if direct else "image_staged_slices") > 0
        known = np.zeros(net.n_neurons, bool)
        known[h.owned] = True
        for p in range(n_shards):
            if p != r:
                known[h.halo_needs(p)] = True
        st = parity.pull_state(h, net)
        for name in ("current_voltage", "is_spiking", "last_firing_time"):
            assert np.array_equal(parity.bits(st[name][known]), parity.bits(net[name][known])), (name, r)
        cells = h.cells_read()
        for name in ("st_last_firing_time", "st_seed"):
            assert np.array_equal(parity.bits(st[name][cells]), parity.bits(net[name][cells])), (name, r)
        assert np.array_equal(parity.bits(st["w_value"][h.owned]), parity.bits(net["w_value"][h.owned]))
        h.close()


def test_direct_form_needs_the_fast_step(snn):
    """weight updates between the unpack and the cells: the ordinary protocol, whatever the option says"""
    import torch
    from snn_amd import parallel
    net = c5_structure(8)
    net["do_plasticity"] = 1
    handles = [parity.device_from_oracle(snn, net, shard=(r, 2), csr=True) for r in range(2)]
    for h in handles:
        h.set_option("halo_direct", 2)
    parallel.LocalExchange(handles, torch.device("cuda", 0), halo=True)
    run_in_lockstep(handles, [60])
    net.run(60)
    for h in handles:
        assert h.stat("halo_direct_steps") == 0
        st = parity.pull_state(h, net)
        assert np.array_equal(parity.bits(st["current_voltage"][h.owned]), parity.bits(net["current_voltage"][h.owned]))
        parity.assert_graph_equal(net, h)
        h.close()


@pytest.mark.timeout(600)
def test_one_way_coupling_and_uncoupled_shards(snn):
    """Edge cases of the plan: a shard that only sends (no halo of its own), one that only receives (no border slices) and --
    a second network -- shards that trade nothing at all."""
    import torch
    import oracle_binding as ob
    from snn_amd import parallel
    for coupled in (True, False):
        lay = parity.Layout([(0, 8, 8), (1, 8, 8)], [(2, 8, 8)])
        net = parity.make_oracle(lay, st_kind=ob.ST_POISSON)
        n = 64
        net["current_voltage"] = ob.uniform_array(21, 2 * n, -65.0, 30.0)
        net["gap_conductance"] = 10.0
        net["st_chance_of_firing"] = 0.02
        net["st_seed"] = np.arange(3, 3 + n, dtype=np.uint32)
        conn = net["connections"]
        for k in range(2):
            for i in range(n):
                conn[k * n + (i + 1) % n, k * n + i] = 1                 # a ring inside each lattice
                conn[k * n + (i + 7) % n, k * n + i] = 1
        for i in range(n):
            conn[2 * n + i, i] = 1                                       # Poisson cells drive lattice 0
            if coupled:
                conn[i, n + (i * 5) % n] = 1                             # lattice 0 -> lattice 1, never back
        net["weights"][...] = conn * np.float32(1.5)
        handles = [parity.device_from_oracle(snn, net, shard=(r, 2), csr=True) for r in range(2)]     # shard r = lattice r
        for h in handles:
            h.set_option("halo_direct", 2)
        parallel.LocalExchange(handles, torch.device("cuda", 0), halo=True)
        plans = [h.exchange_plan() for h in handles]
        assert int(plans[0]["recv_words"]) == 0 and int(plans[1]["send_words"]) == 0
        assert (int(plans[1]["recv_words"]) > 0) == coupled
        run_in_lockstep(handles, [150, 50])
        net.run(200, spike_history=True)
        assert net.spike_history[:, :n].sum() > 3
        for r, h in enumerate(handles):
            st = parity.pull_state(h, net)
            known = np.zeros(2 * n, bool)
            known[h.owned] = True
            known[h.halo_needs(1 - r)] = True
            assert known.sum() == (n if r == 0 or not coupled else 2 * n)
            for name in ("current_voltage", "is_spiking", "last_firing_time"):
                assert np.array_equal(parity.bits(st[name][known]), parity.bits(net[name][known])), (name, r, coupled)
            h.close()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("seed", list(range(int(os.environ.get("SNN_RANDOM_SEEDS_DIRECT", "12")))))     # (env: a longer campaign)
def test_direct_halo_runs_on_random_sparse_networks(snn, seed):
    """random sparse electrical networks (two lattices of unequal size, Poisson cells, rows without inputs or outputs), a random
    number of contiguous shards, random splits of the run"""
    import torch
    from snn_amd import parallel
    from test_gpu_csr import random_sparse_net
    rng = np.random.default_rng(100 + seed)
    net = random_sparse_net(False, density=float(rng.choice([0.02, 0.05, 0.15])), seed=40 + seed)
    net["do_plasticity"] = 0
    g = int(rng.integers(2, 6))
    handles = [parity.device_from_oracle(snn, net, shard=(r, g), csr=True) for r in range(g)]
    for h in handles:
        h.set_option("halo_direct", 2)
        h.set_option("cells_in_step", int(rng.integers(0, 2)))
    ex = parallel.LocalExchange(handles, torch.device("cuda", 0), halo=True)
    calls = [int(x) for x in rng.integers(120, 260, size=3)]
    run_in_lockstep(handles, calls[:2])
    ex.step()
    run_in_lockstep(handles, calls[2:])
    steps = sum(calls) + 1
    net.run(steps, spike_history=True)
    assert net.spike_history.sum() > 0
    for r, h in enumerate(handles):
        if h.owned.size:
            assert h.stat("halo_direct_steps") == sum(calls)
        known = np.zeros(net.n_neurons, bool)
        known[h.owned] = True
        for p in range(g):
            if p != r:
                known[h.halo_needs(p)] = True
        st = parity.pull_state(h, net)
        for name in ("current_voltage", "is_spiking", "last_firing_time"):
            assert np.array_equal(parity.bits(st[name][known]), parity.bits(net[name][known])), (name, r)
        cells = h.cells_read()
        for name in ("st_last_firing_time", "st_seed"):
            assert np.array_equal(parity.bits(st[name][cells]), parity.bits(net[name][cells])), (name, r)
        assert h.clock == steps
        h.close()
