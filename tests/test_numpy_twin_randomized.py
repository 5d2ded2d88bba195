"""The second CPU witness beyond the golden cases: random small networks of every built-in neuron model -- electrical,
chemical or both, every built-in transmitter / receptor kinetics, Poisson / Rate / Preset cells with either refractoriness,
two lattices with their own STDP parameters -- stepped by the C oracle and by the numpy restatement (tests/numpy_net.py:
host libm, no shared code), bit for bit."""
import numpy as np
import pytest

import numpy_net
import oracle_binding as ob
import parity

MODELS = [ob.IZHIKEVICH, ob.LIF, ob.HH, ob.QIF, ob.SIMPLE_LIF, ob.ADAPTIVE_LIF, ob.ADAPTIVE_EXP_LIF, ob.LEAKY_IZHIKEVICH]


def random_net(seed, model):
    rng = np.random.default_rng(seed)
    electrical, chemical = [(True, False), (False, True), (True, True)][seed % 3]
    st_kind = [ob.ST_NONE, ob.ST_POISSON, ob.ST_RATE, ob.ST_PRESET][(seed // 3) % 4]
    nt_kind = [ob.NT_APPROX, ob.NT_DESTEXHE, ob.NT_DISCRETE_SPIKE, ob.NT_EXPONENTIAL_DECAY][(seed // 2) % 4]
    rc_kind = [ob.RC_APPROX, ob.RC_DESTEXHE, ob.RC_EXPONENTIAL_DECAY][(seed // 5) % 3]
    lay = parity.Layout([(0, 3, int(rng.integers(2, 6))), (2, int(rng.integers(1, 4)), 4)],
                        [(1, 2, int(rng.integers(1, 4)))] if st_kind != ob.ST_NONE else [])
    net = parity.make_oracle(lay, model=model, st_kind=st_kind, nt_kind=nt_kind, rc_kind=rc_kind,
                             electrical=electrical, chemical=chemical)
    nn, nc, nt = net.n_neurons, net.n_cells, net.n_tot
    lo, hi = {ob.IZHIKEVICH: (-65, 30), ob.LEAKY_IZHIKEVICH: (-65, 30), ob.HH: (-70, 10)}.get(model, (-70, -54.5))
    net["current_voltage"] = rng.uniform(lo, hi, nn).astype(np.float32)
    net["gap_conductance"] = rng.uniform(1.0, 10.0, nn).astype(np.float32)
    if model in (ob.LIF, ob.QIF, ob.ADAPTIVE_LIF, ob.ADAPTIVE_EXP_LIF):
        net["tref"] = rng.uniform(0.2, 1.5, nn).astype(np.float32)
    if model == ob.LEAKY_IZHIKEVICH:
        net["w_value"] = rng.uniform(0.2, 1.0, nn).astype(np.float32)
    if model == ob.HH:
        net["gap_conductance"] = rng.uniform(0.05, 0.5, nn).astype(np.float32)
    if model == ob.SIMPLE_LIF:
        net["slif_g"] = 0.4
        net["slif_e"] = -76.0
    conn = rng.random((nt, nn)) < 0.6
    conn[np.arange(nn), np.arange(nn)] = False
    net["connections"][...] = conn
    net["weights"][...] = np.where(conn, rng.uniform(-0.5, 1.5, (nt, nn)), 0.0).astype(np.float32)
    net["nt_flags"][...] = rng.random((nn, 3)) < 0.7
    net["rc_flags"][...] = rng.random((nn, 3)) < 0.7
    net["nt_t"][...] = rng.uniform(0.0, 0.3, (nn, 3)).astype(np.float32)
    net["rc_g"][...] = rng.uniform(0.2, 1.5, (nn, 3)).astype(np.float32)
    if nc:
        net["st_nt_flags"][...] = rng.random((nc, 3)) < 0.7
        net["st_chance_of_firing"] = rng.uniform(0.01, 0.1, nc).astype(np.float32)
        net["st_seed"] = rng.integers(1, 2 ** 32, nc, dtype=np.uint32)
        net["st_rate"] = rng.uniform(0.5, 4.0, nc).astype(np.float32)
        net["st_refractoriness"] = rng.integers(0, 2, nc)
        net["st_k"] = rng.uniform(50.0, 10000.0, nc).astype(np.float32)
        if st_kind == ob.ST_PRESET:
            net.set_firing_times([sorted(rng.uniform(0.5, 6.0, int(rng.integers(0, 4))).tolist()) for _ in range(nc)])
    net["do_plasticity"] = [1, seed % 2]
    net["stdp_a_plus"][1] = 1.5
    net["stdp_tau_minus"][0] = 3.0
    return net


@pytest.mark.parametrize("model", MODELS)
@pytest.mark.parametrize("seed", range(12))
def test_oracle_and_numpy_restatement_agree_on_random_networks(model, seed):
    net = random_net(100 * model + seed, model)
    twin = numpy_net.NumpyNet(random_net(100 * model + seed, model))
    steps = 900 if model == ob.HH else 150        # dt = 0.01 for Hodgkin-Huxley: an action potential takes hundreds of steps
    with np.errstate(all="ignore"):
        net.run(steps, voltage_history=True, spike_history=True, st_voltage_history=bool(net.n_cells))
        twin.run(steps, st_voltage_history=bool(net.n_cells))
    assert np.array_equal(twin.spike_history, net.spike_history)
    assert np.array_equal(parity.bits(twin.voltage_history), parity.bits(net.voltage_history))
    names = ["current_voltage", "last_firing_time", "is_spiking", "weights", "nt_t", "rc_r", "rc_current", "w_value",
             "refractory_count", "m_state", "h_state", "n_state", "was_increasing"]
    if net.n_cells:
        names += ["st_last_firing_time", "st_seed", "st_step", "st_counter", "st_nt_t", "st_current_voltage"]
    for k in names:
        assert np.array_equal(parity.bits(twin[k]), parity.bits(net[k])), k
    cases, spikes = SPIKES.get(model, (0, 0))
    SPIKES[model] = (cases + 1, spikes + int(net.spike_history.sum()))


SPIKES = {}


def test_every_model_spiked_somewhere():
    """(runs after the parametrised cases) the comparison above is not one of silent networks"""
    # (under pytest-xdist a worker sees only some of the cases: the check needs every seed of every model in this process)
    if len(SPIKES) == len(MODELS) and all(cases == 12 for cases, _ in SPIKES.values()):
        assert all(spikes > 0 for _, spikes in SPIKES.values()), SPIKES
