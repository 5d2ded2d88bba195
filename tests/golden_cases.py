"""Golden cases shared by tests/golden/make_golden.py (writes the fixtures with the oracle), the CPU test
that pins the oracle to them, and the GPU test that holds the HIP stepper to the SAME committed vectors.
The reference holds no golden vectors for this path (every test draws from thread_rng), so
these are produced here by the pinned oracle; inputs are fully specified by the builders below."""
import numpy as np

import oracle_binding as ob
import parity


def izh_4x4_ones():
    net = parity.make_oracle(parity.Layout([(0, 4, 4)]))
    net["gap_conductance"] = 10.0
    net["current_voltage"] = ob.uniform_array(1, 16, -65.0, 30.0)
    net.connect_all_to_all(1.0)
    return net, 600


def izh_4x4_random():
    net = parity.make_oracle(parity.Layout([(0, 4, 4)]))
    net["gap_conductance"] = 10.0
    net["current_voltage"] = ob.uniform_array(1, 16, -65.0, 30.0)
    net.fill_graph(2, 0.5, 1.5)
    return net, 600


def izh_32x32_random():
    net = parity.make_oracle(parity.Layout([(0, 32, 32)]))
    net["gap_conductance"] = 10.0
    net["current_voltage"] = ob.uniform_array(1, 1024, -65.0, 30.0)
    net.fill_graph(2, 0.5, 1.5)
    return net, 400


def stdp_3_neurons():
    net = parity.make_oracle(parity.Layout([(0, 1, 3)]))
    net["gap_conductance"] = 10.0
    net["current_voltage"] = np.array([25.0, -40.0, 5.0], np.float32)
    net["a"] = np.array([0.02, 0.1, 0.05], np.float32)
    net.connect_all_to_all(1.0)
    net["do_plasticity"] = 1
    return net, 1500


def hh_pair():
    net = parity.make_oracle(parity.Layout([(0, 1, 2)]), model=ob.HH)
    net["current_voltage"] = np.array([-65.0, -20.0], np.float32)
    net["gap_conductance"] = np.array([0.5, 0.1], np.float32)
    net.connect_all_to_all(1.0)
    return net, 2000


def ampa_pair():
    net = parity.make_oracle(parity.Layout([(0, 1, 2)]), chemical=True)
    net["current_voltage"] = np.array([29.5, -65.0], np.float32)
    net["gap_conductance"] = 4.0
    net["nt_flags"][:, 0] = 1
    net["rc_flags"][:, 0] = 1
    net["rc_g"][:, 0] = 2.0
    net.connect_all_to_all(1.0)
    return net, 1200


def spike_trains_poisson():
    net = parity.make_oracle(parity.Layout([(1, 1, 1)], [(0, 4, 4)]), st_kind=ob.ST_POISSON)
    net["st_seed"] = np.arange(1, 17, dtype=np.uint32)                 # seeds 1..16
    net["st_chance_of_firing"] = 0.02
    net["connections"][1:, 0] = 1
    net["weights"][1:, 0] = 1.0
    return net, 1000


def spike_trains_rate():
    net = parity.make_oracle(parity.Layout([(1, 1, 1)], [(0, 2, 3)]), st_kind=ob.ST_RATE)
    net["st_rate"] = np.array([0.0, 1.0, 2.5, 3.0, 7.7, 10.0], np.float32)
    net["connections"][1:, 0] = 1
    net["weights"][1:, 0] = 1.0
    return net, 1000


CASES = {f.__name__: f for f in (izh_4x4_ones, izh_4x4_random, izh_32x32_random, stdp_3_neurons, hh_pair, ampa_pair,
                                 spike_trains_poisson, spike_trains_rate)}

EXTRA = {"hh_pair": ("m_state", "h_state", "n_state"), "stdp_3_neurons": ("weights",),
         "ampa_pair": ("nt_t", "rc_r", "rc_current"), "spike_trains_poisson": ("st_seed", "st_last_firing_time"),
         "spike_trains_rate": ("st_step", "st_last_firing_time")}


def outputs(name, net, steps):
    """What a fixture stores: raster (bit-packed), final state, and the voltage trace (full for small cases,
    every 8th step for 32x32)."""
    stride = 8 if net.n_neurons > 64 else 1
    out = {"steps": np.int64(steps), "trace_stride": np.int64(stride),
           "raster": np.packbits(net.spike_history, axis=1),
           "voltage_trace": net.voltage_history[::stride].copy(),
           "final_voltage": net["current_voltage"].copy(),
           "last_firing_time": net["last_firing_time"].copy()}
    if net.n_cells:
        out["st_voltage_spikes"] = np.packbits(net.st_voltage_history > 0, axis=1)
    for k in EXTRA.get(name, ()):
        out[k] = net[k].copy()
    return out
