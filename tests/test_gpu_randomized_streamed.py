"""Random networks at sizes whose dense matrix is STREAMED (over 64 MB: the 2- and 4-columns-per-lane shapes of the input
pass, STDP riding on it or scattered, shard handles overlapping their own rows with the exchange): ragged populations (no
multiple of 64 or 256), random masks, electrical and / or chemical synapses, spike-train cells, random tuning switches.
Against the oracle, bit for bit.  SNN_RANDOM_SEEDS_STREAMED=n widens the sweep (the suite keeps 6)."""
import os

import numpy as np
import pytest

import oracle_binding as ob
import parity

pytestmark = pytest.mark.gpu

SWITCHES = {"defer_stdp": (0, 1, 2, 3), "uniform_params": (0, 1), "update_packs": (0, 1), "input_shape": (0, 1, 2), "fused_step": (0, 1)}


def draw(seed):
    rng = np.random.default_rng(7000 + seed)
    model = [ob.IZHIKEVICH, ob.LIF, ob.HH, ob.IZHIKEVICH][seed % 4]
    n_total = int(rng.integers(4200, 6400))
    split = int(rng.integers(1, 3))
    sizes = [n_total] if split == 1 else [n_total // 3, n_total - n_total // 3]
    lattices = []
    for i, n in enumerate(sizes):
        rows = int(rng.integers(30, 70))
        lattices.append((2 * i + 1, rows, max(1, n // rows)))
    st_kind = [ob.ST_NONE, ob.ST_POISSON, ob.ST_RATE][int(rng.integers(0, 3))]
    st_lattices = [(50, int(rng.integers(3, 12)), int(rng.integers(3, 12)))] if st_kind != ob.ST_NONE else []
    electrical, chemical = [(True, False), (True, True), (False, True)][int(rng.integers(0, 3))]
    lay = parity.Layout(lattices, st_lattices)
    net = parity.make_oracle(lay, model=model, st_kind=st_kind, electrical=electrical, chemical=chemical,
                             nt_kind=int(rng.integers(0, 2)), rc_kind=int(rng.integers(0, 2)))
    nn, nc = net.n_neurons, net.n_cells
    lo, hi = {ob.IZHIKEVICH: (-65, 30), ob.LIF: (-80, -50), ob.HH: (-75, -40)}[model]
    net["current_voltage"] = ob.uniform_array(seed, nn, lo, hi)
    net["gap_conductance"] = ob.uniform_array(seed + 1, nn, 0.5, 12.0)
    if model == ob.LIF:
        net["tref"] = ob.uniform_array(seed + 2, nn, 0.2, 2.0)
        net["tau_m"] = 10.0
    net["nt_flags"][...] = rng.random((nn, 3)) < np.array([0.9, 0.3, 0.0])[None, :] if seed % 2 else rng.random((nn, 3)) < 0.5
    net["rc_flags"][...] = rng.random((nn, 3)) < 0.5
    net["nt_t"][...] = rng.random((nn, 3)).astype(np.float32) * net["nt_flags"]
    if nc:
        net["st_nt_flags"][...] = rng.random((nc, 3)) < 0.5
        net["st_seed"] = rng.integers(1, 2**32 - 1, nc, dtype=np.uint32)
        net["st_chance_of_firing"] = ob.uniform_array(seed + 4, nc, 0.0, 0.2)
        net["st_rate"] = ob.uniform_array(seed + 5, nc, 0.0, 3.0)
    net.fill_graph(seed + 6, -0.5, 2.0, with_diagonal=bool(rng.integers(0, 2)))
    density = float(rng.choice([0.02, 0.3, 0.9, 1.0]))
    if density < 1.0:
        net["connections"][...] &= (rng.random(net["connections"].shape, dtype=np.float32) < density)
        net["weights"][...] *= net["connections"]
    plastic = bool(rng.integers(0, 2))
    net["do_plasticity"] = int(plastic)
    dt = 0.01 if model == ob.HH else 0.1
    net["dt"] = dt
    net["st_dt"] = dt
    net["stdp_dt"] = dt
    net.n_threads = 8
    plan = dict(shards=int(rng.choice([1, 1, 2, 3])), steps=int(rng.integers(10, 22)), calls=int(rng.integers(1, 3)),
                csr=bool(density <= 0.3 and rng.integers(0, 2)))
    sw = np.random.default_rng(90_000 + seed)
    plan["switches"] = {name: int(sw.choice(values)) for name, values in SWITCHES.items() if sw.integers(0, 2)}
    return net, plan


def threaded(seed):
    return draw(seed)[1]["shards"] > 1 and seed % 2 == 1


@pytest.mark.timeout(900)
@pytest.mark.parametrize("seed", [pytest.param(s, marks=pytest.mark.emulated_ranks) if threaded(s) else s
                                  for s in range(int(os.environ.get("SNN_RANDOM_SEEDS_STREAMED", "6")))])
def test_random_streamed_network(snn, seed):
    import torch
    from snn_amd import parallel
    net, plan = draw(seed)
    steps, g = plan["steps"], plan["shards"]
    if g == 1:
        dn = parity.device_from_oracle(snn, net, csr=plan["csr"])
        for name, value in plan["switches"].items():
            dn.set_option(name, value)
        dn.set_history(voltage=True, spikes=True)
        done = 0
        for c in range(plan["calls"]):
            k = steps // plan["calls"] if c < plan["calls"] - 1 else steps - done
            dn.run(k)
            done += k
        net.run(steps, voltage_history=True, spike_history=True)
        rng = net.layout.ranges()
        for i, _, _ in net.layout.lattices:
            first, count, _ = rng[i]
            assert np.array_equal(dn.spike_history(i), net.spike_history[:, first:first + count])
            assert np.array_equal(parity.bits(dn.voltage_history(i)), parity.bits(net.voltage_history[:, first:first + count]))
        parity.assert_state_equal(net, parity.pull_state(dn, net))
        parity.assert_graph_equal(net, dn)
        dn.close()
        return
    handles = [parity.device_from_oracle(snn, net, shard=(r, g), csr=plan["csr"]) for r in range(g)]
    for h in handles:
        for name, value in plan["switches"].items():
            h.set_option(name, value)
    if seed % 2:
        # the library's own loop (snn_run_sharded) with one host thread per rank: the own-rows chunks of the streamed input pass
        # are enqueued before the loop waits for the collective
        from test_gpu_library_loop_threads import run_ranks
        tc = parallel.ThreadCollectives(g, torch.device("cuda", 0))
        try:
            run_ranks(handles, tc, [steps // 2, steps - steps // 2] if plan["calls"] > 1 else [steps])
        finally:
            tc.close()
    else:
        ex = parallel.LocalExchange(handles, torch.device("cuda", 0), halo=plan["csr"])
        for _ in range(steps):
            for h in handles:
                h.step_begin_local()
                h.step_begin()
            ex.exchange()
            for h in handles:
                h.step_end()
    net.run(steps)
    for h in handles:
        st = parity.pull_state(h, net)
        parity.assert_shard_view_equal(h, st, net)
        if h.csr:
            parity.assert_graph_equal(net, h)
        else:
            b, e = h.post_begin, h.post_end
            w, _ = h.get_graph_rows(0, net.n_tot)
            ow = np.where(net["connections"] != 0, net["weights"], np.float32(0))
            assert np.array_equal(parity.bits(w[:, b:e]), parity.bits(ow[:, b:e]))
        h.close()
