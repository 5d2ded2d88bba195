"""Randomised differential test: many small networks drawn from a seeded generator -- model, kinetics,
synapse kinds, lattice shapes, connectivity density, spike-train inputs, plasticity, graph form (dense / sparse),
whole or sharded stepping, run split into several calls -- each compared bit for bit with the oracle."""
import numpy as np
import pytest

import oracle_binding as ob
import parity

pytestmark = pytest.mark.gpu

MODELS = [ob.IZHIKEVICH, ob.LIF, ob.HH, ob.QIF, ob.SIMPLE_LIF, ob.ADAPTIVE_LIF, ob.ADAPTIVE_EXP_LIF, ob.LEAKY_IZHIKEVICH]


def draw(seed):
    rng = np.random.default_rng(seed)
    model = MODELS[seed % len(MODELS)]
    n_lat = int(rng.integers(1, 4))
    lattices = [(int(i * 2 + rng.integers(0, 2)), int(rng.integers(1, 9)), int(rng.integers(1, 12))) for i in range(n_lat)]
    st_kind = [ob.ST_NONE, ob.ST_POISSON, ob.ST_RATE, ob.ST_PRESET][int(rng.integers(0, 4))]
    st_lattices = []
    if st_kind != ob.ST_NONE:
        st_lattices = [(100 + i, int(rng.integers(1, 6)), int(rng.integers(1, 8))) for i in range(int(rng.integers(1, 3)))]
    electrical, chemical = [(True, False), (True, True), (False, True)][int(rng.integers(0, 3))]
    lay = parity.Layout(lattices, st_lattices)
    net = parity.make_oracle(lay, model=model, st_kind=st_kind, electrical=electrical, chemical=chemical,
                             nt_kind=int(rng.integers(0, 4)), rc_kind=int(rng.integers(0, 3)))
    nn, nc = net.n_neurons, net.n_cells
    lo, hi = {ob.IZHIKEVICH: (-65, 30), ob.LIF: (-80, -50), ob.HH: (-75, -40), ob.QIF: (-75, -56),
              ob.SIMPLE_LIF: (-75, -56), ob.ADAPTIVE_LIF: (-75, -56), ob.ADAPTIVE_EXP_LIF: (-75, -56),
              ob.LEAKY_IZHIKEVICH: (-65, 30)}[model]
    net["current_voltage"] = ob.uniform_array(seed, nn, lo, hi)
    net["gap_conductance"] = ob.uniform_array(seed + 1, nn, 0.5, 12.0)
    if model == ob.SIMPLE_LIF:
        net["slif_g"] = 0.3
        net["slif_e"] = -76.0
    if model in (ob.LIF, ob.QIF, ob.ADAPTIVE_LIF, ob.ADAPTIVE_EXP_LIF):
        net["tref"] = ob.uniform_array(seed + 2, nn, 0.2, 2.0)
        net["tau_m"] = 10.0
    if model in (ob.ADAPTIVE_LIF, ob.ADAPTIVE_EXP_LIF):
        net["leak_constant"] = 1.0
        net["c_m"] = 1.0
        net["v_reset"] = -73.0
        net["adp_beta"] = ob.uniform_array(seed + 7, nn, 0.5, 4.0)
    if model == ob.ADAPTIVE_EXP_LIF:
        net["slope_factor"] = ob.uniform_array(seed + 8, nn, 0.5, 3.0)
    if model == ob.LEAKY_IZHIKEVICH:
        net["w_value"] = ob.uniform_array(seed + 7, nn, 0.0, 1.0)
    net["nt_flags"][...] = rng.random((nn, 3)) < 0.5
    net["rc_flags"][...] = rng.random((nn, 3)) < 0.5
    net["rc_g"][...] *= ob.uniform_array(seed + 3, 3 * nn, 0.5, 3.0).reshape(nn, 3)
    net["nt_t"][...] = rng.random((nn, 3)).astype(np.float32) * net["nt_flags"]
    if nc:
        net["st_nt_flags"][...] = rng.random((nc, 3)) < 0.5
        net["st_seed"] = rng.integers(1, 2**32 - 1, nc, dtype=np.uint32)
        net["st_chance_of_firing"] = ob.uniform_array(seed + 4, nc, 0.0, 0.08)
        net["st_rate"] = ob.uniform_array(seed + 5, nc, 0.0, 6.0)
        if st_kind == ob.ST_PRESET:
            net.set_firing_times([list(rng.uniform(0.3, 5.0, int(k))) for k in rng.integers(0, 4, nc)])
    net.fill_graph(seed + 6, -0.5, 2.0, with_diagonal=bool(rng.integers(0, 2)))
    density = float(rng.choice([0.05, 0.3, 0.8, 1.0]))
    net["connections"][...] &= (rng.random(net["connections"].shape) < density)
    net["weights"][...] *= net["connections"]
    for slot in range(len(lattices)):
        net["do_plasticity"][slot] = int(rng.integers(0, 2))
        net["stdp_a_plus"][slot] = float(rng.uniform(0.5, 2.5))
        net["stdp_tau_minus"][slot] = float(rng.uniform(2.0, 6.0))
    dt = 0.01 if model == ob.HH else float(rng.choice([0.05, 0.1, 0.2]))
    net["dt"] = dt
    net["st_dt"] = dt
    net["stdp_dt"] = dt
    plan = dict(csr=bool(rng.integers(0, 2)), shards=int(rng.choice([1, 1, 2, 3])),
                steps=int(rng.integers(80, 260)), calls=int(rng.integers(1, 4)))
    return net, plan


@pytest.mark.parametrize("seed", list(range(48)))
def test_random_network(snn, seed):
    import torch
    from snn_amd import parallel
    net, plan = draw(1000 + seed)
    steps = plan["steps"]
    if plan["shards"] == 1:
        dn = parity.device_from_oracle(snn, net, csr=plan["csr"])
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
        assert dn.clock == net.clock
        dn.close()
        return
    g = plan["shards"]
    handles = [parity.device_from_oracle(snn, net, shard=(r, g), csr=plan["csr"]) for r in range(g)]
    bufs = [parallel.exchange_tensor(h, torch.device("cuda", 0)) for h in handles]
    block = bufs[0].numel() // g
    for _ in range(steps):
        for h in handles:
            h.step_begin_local()
            h.step_begin()
        for r in range(g):
            for o in range(g):
                if o != r:
                    bufs[o][r * block:(r + 1) * block].copy_(bufs[r][r * block:(r + 1) * block])
        torch.cuda.synchronize()
        for h in handles:
            h.step_end()
    net.run(steps)
    for h in handles:
        st = parity.pull_state(h, net)
        for name in ("current_voltage", "is_spiking", "last_firing_time", "nt_t"):
            assert np.array_equal(parity.bits(st[name]), parity.bits(net[name])), name
        b, e = h.post_begin, h.post_end
        for name in ("rc_r", "rc_current"):
            assert np.array_equal(parity.bits(st[name][b:e]), parity.bits(net[name][b:e])), name
        if h.csr:
            parity.assert_graph_equal(net, h)
        elif net.n_neurons and net.n_tot:
            w, c = h.get_graph_rows(0, net.n_tot)
            ow = np.where(net["connections"] != 0, net["weights"], np.float32(0))
            assert np.array_equal(parity.bits(w[:, b:e]), parity.bits(ow[:, b:e]))
        h.close()
