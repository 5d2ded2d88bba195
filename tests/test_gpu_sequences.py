"""Random SEQUENCES of C-ABI calls on one handle against the same sequence on the oracle: runs of random length interleaved with
writes behind the stepper's back (voltages, the cells' firing times), tuning switches, reads of weights and state (which flush
pending deferred updates), plasticity and synapse-kind toggles and reset_timing.  Exercises the handle's bookkeeping between run
calls: the spike-train view, the shadows of the exchange buffer, pending weight updates, the exchange-independent caches.
SNN_RANDOM_SEEDS_SEQUENCES=n widens the sweep (the suite keeps 40)."""
import os

import numpy as np
import pytest

import checkpoint
import oracle_binding as ob
import parity
from test_gpu_randomized import SWITCHES, draw, make_handle

pytestmark = pytest.mark.gpu


def usable(seed):
    net, plan = draw(1000 + seed)
    return plan["rewards"] is None and not net["plasticity_kind"].any() and net.n_neurons > 0


SEEDS = [s for s in range(int(os.environ.get("SNN_RANDOM_SEEDS_SEQUENCES", "90"))) if usable(s)]


class _Absent:
    """stands in for the side a replay leaves out: every call is accepted and does nothing"""
    clock = 0

    def __getattr__(self, name):
        return lambda *a, **k: None


def play(snn, seed, device=True, oracle=True, names=()):
    """One random sequence of calls (a function of the seed alone) on a device handle and on the oracle; `device` / `oracle` False
    replays it on ONE side only (the other side's calls are dropped, the oracle's container still supplies the numbers the
    sequence draws from).  Returns (device state, oracle state, log); compares on the way only when both sides run."""
    net, plan = draw(1000 + seed)
    rng = np.random.default_rng(300_000 + seed)
    both = device and oracle
    dn = make_handle(snn, net, plan) if device else _Absent()
    real_run = net.run

    def oracle_run(k, **kw):
        if oracle:
            return real_run(k, **kw)
        net.clock += k                                       # the sequence draws firing times below the clock
    net.run = oracle_run
    ranges = net.layout.ranges()
    lattices = [i for i, _, _ in net.layout.lattices if ranges[i][1]]
    cells = [i for i, _, _ in net.layout.st_lattices if ranges[i][1]]
    lo, hi = float(net["current_voltage"].min()) - 1.0, float(net["current_voltage"].max()) + 1.0
    log = []
    # two seeds in three: device and oracle compared at every run-call boundary, the diverging call executed again from a checkpoint
    tr = checkpoint.Tracker(snn, dn, net, plan, f"sequence-{seed}", log=log, enabled=both and seed % 3 != 1)
    for _ in range(int(rng.integers(4, 10))):
        op = int(rng.integers(0, 14))
        if op <= 2:
            k = int(rng.integers(1, 70))
            tr.run(k) if tr.enabled else (dn.run(k), net.run(k))
            log.append(("run", k))
        elif op == 3:
            i = int(rng.choice(lattices))
            first, count, _ = ranges[i]
            v = rng.uniform(lo, hi, count).astype(np.float32)
            dn.set_attr(i, "current_voltage", v)
            net["current_voltage"][first:first + count] = v
            log.append(("voltage", i))
        elif op == 4 and cells:
            i = int(rng.choice(cells))
            first, count, _ = ranges[i]
            t = np.where(rng.random(count) < 0.5, -1, rng.integers(0, max(1, net.clock + 1), count)).astype(np.int32)
            dn.set_attr(i, "last_firing_time", t)
            net["st_last_firing_time"][first:first + count] = t
            log.append(("cell firing times", i))
        elif op == 5:
            name = str(rng.choice(list(SWITCHES)))
            value = int(rng.choice(SWITCHES[name]))
            dn.set_option(name, value)
            tr.options[name] = value
            log.append(("switch", name, value))
        elif op == 6:
            check_state = bool(rng.integers(0, 2))
            if both:
                parity.assert_graph_equal(net, dn)
                if check_state:
                    parity.assert_state_equal(net, parity.pull_state(dn, net))
            elif device:
                dn.get_graph_rows(0, net.n_tot) if not plan["csr"] else dn.get_graph_csr()      # (a read flushes pending updates)
                if check_state:
                    parity.pull_state(dn, net)
            log.append(("read",))
        elif op == 7:
            slot = int(rng.integers(0, len(net.layout.lattices)))
            i = net.layout.lattices[slot][0]
            on = bool(rng.integers(0, 2))
            dn.set_plasticity(i, float(net["stdp_a_plus"][slot]), float(net["stdp_a_minus"][slot]), float(net["stdp_tau_plus"][slot]),
                              float(net["stdp_tau_minus"][slot]), float(net["stdp_dt"][slot]), on)
            net["do_plasticity"][slot] = int(on)
            log.append(("plasticity", i, on))
        elif op == 8:
            dn.reset_timing()
            net.clock = 0                                    # neuron/mod.rs:405-420, 1710-1717
            net["last_firing_time"][...] = -1
            if net.n_cells:
                net["st_last_firing_time"][...] = -1
                net["st_clock"][...] = 0
            log.append(("reset_timing",))
        elif op == 9:
            # parameters and flags rewritten wholesale (uniform-parameter table, static counts, live transmitter types)
            nn = net.n_neurons
            net["gap_conductance"] = float(rng.uniform(0.5, 12.0)) if rng.integers(0, 2) else rng.uniform(0.5, 12.0, nn).astype(np.float32)
            net["nt_flags"][...] = rng.random((nn, 3)) < 0.5
            net["nt_t"][...] = net["nt_t"] * net["nt_flags"]
            net["rc_flags"][...] = rng.random((nn, 3)) < 0.5
            if device:
                parity.push_state(dn, net)
            log.append(("parameters",))
        elif op == 10:
            el, ch = [(True, False), (True, True), (False, True)][int(rng.integers(0, 3))]
            dn.set_synapses(el, ch)
            net.electrical, net.chemical = el, ch
            log.append(("synapses", el, ch))
        elif op == 11 and net.n_tot:
            # the weights of the existing edges rewritten (same mask)
            w = rng.uniform(-0.5, 2.0, net["weights"].shape).astype(np.float32) * net["connections"]
            net["weights"][...] = w
            if not device:
                pass
            elif plan["csr"]:
                dn.set_graph_csr(*parity.csr_for_posts(net, dn.owned))
            else:
                dn.set_graph_rows(0, net["weights"], net["connections"].astype(np.uint32))
            log.append(("weights",))
        elif op == 12 and cells:
            i = int(rng.choice(cells))
            first, count, _ = ranges[i]
            if net.st_kind in (ob.ST_POISSON, ob.ST_BCM_POISSON):
                seeds = rng.integers(1, 2**32 - 1, count, dtype=np.uint32)
                dn.set_attr(i, "seed", seeds)
                net["st_seed"][first:first + count] = seeds
                log.append(("seeds", i))
        elif op == 13:
            # a recorded stretch: histories switched on (strided capture), compared, switched off again
            k, every = int(rng.integers(1, 40)), int(rng.choice([1, 1, 2, 3]))
            dn.reset_history()
            dn.set_history(voltage=True, spikes=True)
            dn.set_history_stride(every)
            dn.run(k)
            net.run(k, voltage_history=True, spike_history=True)
            keep = np.arange(0, k, every)
            for i in (lattices if both else ()):
                first, count, _ = ranges[i]
                assert np.array_equal(dn.spike_history(i), net.spike_history[keep, first:first + count]), f"raster; sequence: {log}"
                assert np.array_equal(parity.bits(dn.voltage_history(i)), parity.bits(net.voltage_history[keep, first:first + count])), f"trace; sequence: {log}"
            dn.set_history(voltage=False, spikes=False)
            dn.set_history_stride(1)
            log.append(("recorded", k, every))
    tr.run(5) if tr.enabled else (dn.run(5), net.run(5))
    dev = parity.pull_state(dn, net) if device else None
    orc = {k: np.array(net[k], copy=True) for k in (dev if dev is not None else names)} if oracle else None
    if both:
        try:
            parity.assert_state_equal(net, dev)
            parity.assert_graph_equal(net, dn)
            assert dn.clock == net.clock
        except AssertionError as e:
            dn.close()
            e.sides = (dev, orc, log)
            raise
    dn.close()
    return dev, orc, log


def same(a, b):
    return all(np.array_equal(parity.bits(a[k]) if a[k].dtype == np.float32 else a[k], parity.bits(b[k]) if b[k].dtype == np.float32 else b[k])
               for k in a)


@pytest.mark.parametrize("seed", SEEDS)
def test_random_call_sequence(snn, seed):
    """On a mismatch BOTH sides are replayed alone, from the seed, in this process: the message says which side the replay
    reproduces -- a device that differs from its own replay points at the stepper, an oracle that does at the checker."""
    try:
        play(snn, seed)
    except AssertionError as e:
        dev, orc, log = getattr(e, "sides", (None, None, []))
        verdict = "no replay (the mismatch came from a comparison inside the sequence)"
        if dev is not None:
            dev2 = play(snn, seed, oracle=False)[0]
            orc2 = play(snn, seed, device=False, names=tuple(dev))[1]
            verdict = (f"replay of the device alone {'REPRODUCES' if same(dev, dev2) else 'DIFFERS FROM'} the failing run's device state; "
                       f"replay of the oracle alone {'REPRODUCES' if same(orc, orc2) else 'DIFFERS FROM'} the failing run's oracle state; "
                       f"the two replays {'AGREE' if same(dev2, orc2) else 'disagree'} with each other")
        raise AssertionError(f"{e}; sequence: {log}; {verdict}") from e


def usable_sharded(seed):
    net, plan = draw(1000 + seed)
    return plan["rewards"] is None and not net["plasticity_kind"].any() and net.n_neurons > 0 and plan["shards"] > 1


SHARDED_SEEDS = [s for s in range(int(os.environ.get("SNN_RANDOM_SEEDS_SEQUENCES", "90")) * 2) if usable_sharded(s)]


@pytest.mark.emulated_ranks
@pytest.mark.timeout(600)
@pytest.mark.parametrize("seed", SHARDED_SEEDS)
def test_random_call_sequence_on_shard_handles(snn, seed):
    """the same idea on G shard handles: stretches of the library's own loop (snn_run_sharded, one host thread per rank, replaced
    collectives) alternate with host-driven steps, and between them the exchange plan is made to change (synapse kinds), state
    is rewritten, switches are flipped -- the plan, its agreement between the ranks and the direct form are rebuilt on the way"""
    import torch
    from snn_amd import parallel
    from test_gpu_library_loop_threads import run_ranks
    net, plan = draw(1000 + seed)
    rng = np.random.default_rng(400_000 + seed)
    g = plan["shards"]
    handles = [make_handle(snn, net, plan, shard=(r, g)) for r in range(g)]
    dev = torch.device("cuda", 0)
    ex = parallel.LocalExchange(handles, dev, halo=plan["csr"])
    ranges = net.layout.ranges()
    lattices = [i for i, _, _ in net.layout.lattices if ranges[i][1]]
    lo, hi = float(net["current_voltage"].min()) - 1.0, float(net["current_voltage"].max()) + 1.0
    log = []
    for _ in range(int(rng.integers(3, 8))):
        op = int(rng.integers(0, 8))
        if op <= 1:
            k = int(rng.integers(1, 50))
            tc = parallel.ThreadCollectives(g, dev)
            try:
                run_ranks(handles, tc, [k])
            finally:
                tc.close()
            net.run(k)
            log.append(("library loop", k))
        elif op == 2:
            k = int(rng.integers(1, 20))
            ex.refresh_state()                    # (a synapse kind may have been switched on: the mirrors catch up first)
            for _ in range(k):
                for h in handles:
                    h.step_begin_local()
                    h.step_begin()
                ex.exchange()
                for h in handles:
                    h.step_end()
            net.run(k)
            log.append(("host-driven", k))
        elif op == 3:
            i = int(rng.choice(lattices))
            first, count, _ = ranges[i]
            v = rng.uniform(lo, hi, count).astype(np.float32)
            for h in handles:
                h.set_attr(i, "current_voltage", v)
            net["current_voltage"][first:first + count] = v
            log.append(("voltage", i))
        elif op == 4:
            name = str(rng.choice(list(SWITCHES) + ["halo_direct", "cells_in_step"]))
            value = int(rng.choice(SWITCHES.get(name, (0, 1))))
            for h in handles:
                h.set_option(name, value)
            log.append(("switch", name, value))
        elif op == 5:
            el, ch = [(True, False), (True, True), (False, True)][int(rng.integers(0, 3))]
            for h in handles:
                h.set_synapses(el, ch)
            net.electrical, net.chemical = el, ch
            log.append(("synapses", el, ch))
        elif op == 6:
            slot = int(rng.integers(0, len(net.layout.lattices)))
            i = net.layout.lattices[slot][0]
            on = bool(rng.integers(0, 2))
            for h in handles:
                h.set_plasticity(i, float(net["stdp_a_plus"][slot]), float(net["stdp_a_minus"][slot]), float(net["stdp_tau_plus"][slot]),
                                 float(net["stdp_tau_minus"][slot]), float(net["stdp_dt"][slot]), on)
            net["do_plasticity"][slot] = int(on)
            log.append(("plasticity", i, on))
        elif op == 7:
            for h in handles:
                if h.csr:
                    parity.assert_graph_equal(net, h)
            log.append(("read",))
    tc = parallel.ThreadCollectives(g, dev)
    try:
        run_ranks(handles, tc, [4])
    finally:
        tc.close()
    net.run(4)
    try:
        for h in handles:
            st = parity.pull_state(h, net)
            parity.assert_shard_view_equal(h, st, net)
            assert h.clock == net.clock
            if h.csr:
                parity.assert_graph_equal(net, h)
            elif net.n_tot:
                b, e = h.post_begin, h.post_end
                w, _ = h.get_graph_rows(0, net.n_tot)
                ow = np.where(net["connections"] != 0, net["weights"], np.float32(0))
                assert np.array_equal(parity.bits(w[:, b:e]), parity.bits(ow[:, b:e])), "weights"
    except AssertionError as e:
        raise AssertionError(f"{e}; sequence: {log}") from e
    for h in handles:
        h.close()
