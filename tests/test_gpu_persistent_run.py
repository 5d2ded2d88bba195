"""Many steps in ONE launch (k_run_resident: electrical-only lattices of <= 4096 neurons without plasticity, the shape of
BASELINE configs[0]) against the one-launch-per-step form (option persistent_run 0) and the oracle: bit-identical
voltages, rasters, state and spike totals, for ragged sizes, split run calls, every built-in model, sparse connectivity,
and voltages that leave the range in which an absent edge may be summed as a zero product."""
import os

import numpy as np
import pytest

import oracle_binding as ob
import parity

pytestmark = pytest.mark.gpu


def build(model, rows, cols, seed, density=0.75):
    net = parity.make_oracle(parity.Layout([(3, rows, cols)]), model=model, electrical=True, chemical=False)
    n = net.n_neurons
    lo, hi = {ob.IZHIKEVICH: (-65, 30), ob.LIF: (-80, -50), ob.HH: (-75, -40), ob.QIF: (-75, -56)}.get(model, (-70, -50))
    net["current_voltage"] = ob.uniform_array(seed, n, lo, hi)
    net["gap_conductance"] = ob.uniform_array(seed + 5, n, 2.0, 10.0) if model == ob.IZHIKEVICH else 3.0
    if model in (ob.LIF, ob.QIF):
        net["tref"] = ob.uniform_array(seed + 1, n, 0.3, 1.5)
        net["tau_m"] = 10.0
    rng = np.random.default_rng(seed)
    net.fill_graph(seed + 3, 0.5, 1.5)
    net["connections"][rng.random(net["connections"].shape) >= density] = 0
    net["weights"][...] *= net["connections"]
    net["do_plasticity"] = 0
    return net


def run_device(snn, net, calls, persistent):
    dn = parity.device_from_oracle(snn, net)
    dn.set_option("persistent_run", int(persistent))
    dn.set_history(voltage=True, spikes=True)
    dn.set_reduced_history(spike_counts=True)
    for steps in calls:
        dn.run(steps)
    out = {"state": parity.pull_state(dn, net), "v": dn.voltage_history(3), "s": dn.spike_history(3),
           "counts": dn.spike_counts(3), "launches": dn.stat("persistent_run_launches"), "steps": dn.stat("persistent_run_steps")}
    dn.close()
    return out


def compare(snn, net, calls):
    a = run_device(snn, net, calls, True)
    b = run_device(snn, net, calls, False)
    expected = [c for c in calls if c >= 4]
    assert a["launches"] == len(expected) and a["steps"] == sum(expected) and b["launches"] == 0
    total = sum(calls)
    net.run(total, voltage_history=True, spike_history=True, spike_counts=True)
    for out in (a, b):
        parity.assert_state_equal(net, out["state"])
        assert np.array_equal(out["s"], net.spike_history)
        assert np.array_equal(parity.bits(out["v"]), parity.bits(net.voltage_history))
        assert np.array_equal(out["counts"], net.spike_counts)
    return a


@pytest.mark.parametrize("rows,cols,seed", [(1, 1, 1), (5, 5, 2), (7, 9, 3), (8, 8, 4), (16, 17, 5), (30, 30, 6), (32, 32, 7), (31, 33, 8)])
def test_izhikevich_sizes(snn, rows, cols, seed):
    net = build(ob.IZHIKEVICH, rows, cols, seed)
    a = compare(snn, net, [400, 1, 3, 4, 193])
    assert a["s"].sum() > 0


@pytest.mark.parametrize("rows,cols,seed", [(33, 33, 41), (32, 40, 42), (45, 45, 43), (48, 48, 44), (56, 57, 45), (64, 64, 46)])
def test_several_row_groups_per_tile(snn, rows, cols, seed):
    """1025 .. 4096 neurons: 2 .. 4 row groups of 1024 rows per column tile, the chunk sums of groups 1.. travel to group 0."""
    net = build(ob.IZHIKEVICH, rows, cols, seed, density=0.5)
    a = compare(snn, net, [150, 3, 47])
    assert a["s"].sum() > 0


def test_several_row_groups_generic_update(snn):
    net = build(ob.LIF, 40, 41, 51, density=0.5)
    a = compare(snn, net, [200])
    assert a["s"].sum() > 0


@pytest.mark.parametrize("model", [ob.LIF, ob.HH, ob.QIF])
def test_other_models(snn, model):
    net = build(model, 18, 19, 11 + model)
    a = compare(snn, net, [250, 250] if model != ob.HH else [500, 400])
    assert a["s"].sum() > 0


def test_sparse_connectivity_and_isolated_neurons(snn):
    net = build(ob.IZHIKEVICH, 24, 24, 21, density=0.02)
    net["connections"][:, 5] = 0
    net["connections"][7, :] = 0
    net["weights"][...] *= net["connections"]
    a = compare(snn, net, [600])
    assert a["s"].sum() > 0


def test_voltages_outside_the_zero_product_range(snn):
    """A neuron whose voltage is huge, infinite or NaN reaches only the neurons it has an edge to: steps holding such a
    value take the path with explicit edge selects (an absent edge times an infinite term would be NaN, not zero)."""
    def bad_net(bad):
        net = build(ob.IZHIKEVICH, 12, 12, 31, density=0.03)
        net["connections"][17, :] = 0
        net["connections"][17, :10] = 1               # the bad neuron feeds ten others only
        net["weights"][...] = net["weights"] * 0 + net["connections"]
        v = net["current_voltage"].copy()
        v[17] = bad
        net["current_voltage"] = v
        return net

    for bad in (3.0e20, np.inf, np.nan):
        a = compare(snn, bad_net(bad), [4])
        assert np.isfinite(a["v"][0]).sum() > 100         # after one step: the neurons without an edge from it
        compare(snn, bad_net(bad), [40])


MODELS = [ob.IZHIKEVICH, ob.LIF, ob.HH, ob.QIF, ob.SIMPLE_LIF, ob.ADAPTIVE_LIF, ob.ADAPTIVE_EXP_LIF, ob.LEAKY_IZHIKEVICH]


def draw(seed):
    """1-3 lattices of one model with per-neuron parameters, random density (0 included), negative weights; every other
    network with one or two lattices of Poisson / Rate cells; every third one with chemical synapses (with or without gap
    junctions, random transmitter / receptor flags and kinetics, one row group: <= 1024 rows), the others electrical only."""
    rng = np.random.default_rng(1000 + seed)
    model = MODELS[seed % len(MODELS)]
    chem = seed % 3 == 2
    big = seed % 5 == 0 and not chem                      # some networks beyond one row group
    lattices = []
    for i in range(int(rng.integers(1, 4))):
        hi_side = 26 if big else 14
        lattices.append((int(2 * i + rng.integers(0, 2)), int(rng.integers(1, hi_side)), int(rng.integers(1, hi_side))))
    st_kind, st_lattices = ob.ST_NONE, []
    if seed % 2:                                          # every other network: Poisson or Rate cells as presynaptic rows
        st_kind = ob.ST_POISSON if seed % 4 == 1 else ob.ST_RATE
        st_lattices = [(100 + i, int(rng.integers(1, 12)), int(rng.integers(1, 12))) for i in range(int(rng.integers(1, 3)))]
    electrical = not chem or bool(rng.integers(0, 2))
    kinetics = dict(nt_kind=int(rng.integers(0, 4)), rc_kind=int(rng.integers(0, 3))) if chem else {}
    net = parity.make_oracle(parity.Layout(lattices, st_lattices), model=model, st_kind=st_kind, electrical=electrical, chemical=chem,
                             **kinetics)
    n = net.n_neurons
    net["nt_flags"][...] = 0
    if chem:
        live = rng.random(3) < 0.6                        # transmitter types nobody releases stay off the wire
        net["nt_flags"][...] = (rng.random((n, 3)) < 0.5) & live
        net["rc_flags"][...] = rng.random((n, 3)) < 0.6
        net["rc_g"][...] *= ob.uniform_array(seed + 23, 3 * n, 0.5, 3.0).reshape(n, 3)
        net["nt_t"][...] = rng.random((n, 3)).astype(np.float32)     # also where the flag is off: held, never released
    if net.n_cells:
        nc = net.n_cells
        net["st_nt_flags"][...] = 0
        net["st_seed"] = rng.integers(1, 2**32 - 1, nc, dtype=np.uint32)
        net["st_chance_of_firing"] = ob.uniform_array(seed + 14, nc, 0.0, 0.08)
        net["st_rate"] = ob.uniform_array(seed + 15, nc, 0.0, 6.0)
        net["st_refractoriness"][...] = rng.integers(0, 2, nc)
        net["st_k"] = ob.uniform_array(seed + 16, nc, 20.0, 10000.0)
    lo, hi = {ob.IZHIKEVICH: (-65, 30), ob.LIF: (-80, -50), ob.HH: (-75, -40), ob.LEAKY_IZHIKEVICH: (-65, 30)}.get(model, (-75, -56))
    net["current_voltage"] = ob.uniform_array(seed, n, lo, hi)
    net["gap_conductance"] = ob.uniform_array(seed + 1, n, 0.5, 12.0)
    if model in (ob.IZHIKEVICH, ob.LEAKY_IZHIKEVICH):
        net["a"] = ob.uniform_array(seed + 2, n, 0.01, 0.1)
        net["b"] = ob.uniform_array(seed + 3, n, 0.15, 0.3)
        net["c"] = ob.uniform_array(seed + 4, n, -70.0, -50.0)
        net["d"] = ob.uniform_array(seed + 5, n, 2.0, 9.0)
        net["v_th"] = ob.uniform_array(seed + 6, n, 20.0, 35.0)
    if model == ob.LEAKY_IZHIKEVICH:
        net["w_value"] = ob.uniform_array(seed + 7, n, 0.0, 1.0)
    if model == ob.SIMPLE_LIF:
        net["slif_g"] = 0.3
        net["slif_e"] = -76.0
    if model in (ob.LIF, ob.QIF, ob.ADAPTIVE_LIF, ob.ADAPTIVE_EXP_LIF):
        net["tref"] = ob.uniform_array(seed + 2, n, 0.2, 2.0)
        net["tau_m"] = 10.0
    if model in (ob.ADAPTIVE_LIF, ob.ADAPTIVE_EXP_LIF):
        net["leak_constant"] = 1.0
        net["c_m"] = 1.0
        net["v_reset"] = -73.0
        net["adp_beta"] = ob.uniform_array(seed + 7, n, 0.5, 4.0)
    if model == ob.ADAPTIVE_EXP_LIF:
        net["slope_factor"] = ob.uniform_array(seed + 8, n, 0.5, 3.0)
    net.fill_graph(seed + 9, -0.5, 2.0, with_diagonal=bool(rng.integers(0, 2)))
    density = float(rng.choice([0.0, 0.05, 0.3, 0.8, 1.0]))
    net["connections"][...] &= (rng.random(net["connections"].shape) < density)
    net["weights"][...] *= net["connections"]
    net["do_plasticity"] = 0
    if seed % 14 == 6 and not chem and not big:           # (even seeds carry no cells) STDP inside the run: plastic lattices, their own
        for l in range(len(lattices)):                    # rules, weak coupling so that the neurons keep firing one after the other
            net["do_plasticity"][l] = int(rng.random() < 0.8)
            net["stdp_a_plus"][l] = float(rng.uniform(0.5, 2.5))
            net["stdp_a_minus"][l] = float(rng.uniform(0.5, 2.5))
            net["stdp_tau_plus"][l] = float(rng.uniform(2.0, 6.0))
            net["stdp_tau_minus"][l] = float(rng.uniform(2.0, 6.0))
        net["do_plasticity"][0] = 1
        net["gap_conductance"] = ob.uniform_array(seed + 1, n, 0.2, 1.0)
        if model in (ob.IZHIKEVICH, ob.LEAKY_IZHIKEVICH, ob.QIF):
            net["current_voltage"] = ob.uniform_array(seed, n, -70.0, 19.9)
    calls = [int(c) for c in rng.integers(1, 120, int(rng.integers(1, 4)))]
    return net, calls, bool(rng.integers(0, 2)), bool(rng.integers(0, 2))


def build_chemical(model, rows, cols, seed, types, nt_kind, rc_kind, electrical=True, density=0.75):
    net = parity.make_oracle(parity.Layout([(3, rows, cols)]), model=model, electrical=electrical, chemical=True, nt_kind=nt_kind, rc_kind=rc_kind)
    n = net.n_neurons
    rng = np.random.default_rng(seed)
    lo, hi = {ob.IZHIKEVICH: (-65, 30), ob.LIF: (-80, -50), ob.HH: (-75, -40), ob.QIF: (-75, -56)}.get(model, (-70, -50))
    net["current_voltage"] = ob.uniform_array(seed, n, lo, hi)
    net["gap_conductance"] = ob.uniform_array(seed + 5, n, 2.0, 10.0)
    if model in (ob.LIF, ob.QIF):
        net["tref"] = ob.uniform_array(seed + 1, n, 0.3, 1.5)
        net["tau_m"] = 10.0
    flags = np.zeros((n, 3), bool)
    flags[:, list(types)] = rng.random((n, len(types))) < 0.7
    net["nt_flags"][...] = flags
    net["rc_flags"][...] = rng.random((n, 3)) < 0.7
    net["rc_g"][...] *= ob.uniform_array(seed + 23, 3 * n, 0.5, 2.0).reshape(n, 3)
    net["nt_t"][...] = rng.random((n, 3)).astype(np.float32)
    net.fill_graph(seed + 3, 0.5, 1.5)
    net["connections"][rng.random(net["connections"].shape) >= density] = 0
    net["weights"][...] *= net["connections"]
    net["do_plasticity"] = 0
    return net


@pytest.mark.parametrize("model,rows,cols,types,nt_kind,rc_kind,electrical,seed", [
    (ob.IZHIKEVICH, 4, 4, (0,), 0, 0, True, 61), (ob.IZHIKEVICH, 16, 16, (0,), 0, 0, True, 62), (ob.IZHIKEVICH, 32, 32, (0,), 0, 0, True, 63),
    (ob.IZHIKEVICH, 31, 33, (0, 1, 2), 1, 1, True, 64), (ob.IZHIKEVICH, 20, 20, (1, 2), 2, 2, False, 65),
    (ob.HH, 12, 12, (0,), 1, 1, True, 66), (ob.LIF, 9, 30, (0, 2), 3, 0, True, 67), (ob.QIF, 8, 8, (), 0, 0, True, 68)])
def test_chemical_synapses_in_the_one_launch_run(snn, model, rows, cols, types, nt_kind, rc_kind, electrical, seed):
    """k_run_resident<..., CHEM>: the released concentrations travel as granules of their own next to the voltages, the weighted
    sums of every live transmitter type keep the canonical order -- bit-identical to the one-launch-per-step form and the
    oracle, for every kinetics pair, several live types, no live type at all, and chemical synapses alone."""
    net = build_chemical(model, rows, cols, seed, types, nt_kind, rc_kind, electrical)
    a = compare(snn, net, [150, 2, 5, 43])
    assert a["launches"] == 3


def test_chemical_run_with_cells_and_rollback(snn):
    """cells that release nothing ride along (Poisson rows); a faulted launch is rolled back and repeated per step"""
    lay = parity.Layout([(3, 18, 18)], [(1, 6, 10)])
    net = parity.make_oracle(lay, model=ob.IZHIKEVICH, st_kind=ob.ST_POISSON, electrical=True, chemical=True, nt_kind=0, rc_kind=0)
    n = net.n_neurons
    rng = np.random.default_rng(71)
    net["st_chance_of_firing"] = 0.05
    net["st_nt_flags"][...] = 0
    net["current_voltage"] = ob.uniform_array(71, n, -65, 30)
    net["gap_conductance"] = ob.uniform_array(72, n, 2.0, 10.0)
    net["nt_flags"][...] = rng.random((n, 3)) < 0.6
    net["rc_flags"][...] = rng.random((n, 3)) < 0.6
    net["nt_t"][...] = rng.random((n, 3)).astype(np.float32) * net["nt_flags"]
    net.fill_graph(73, 0.5, 1.5)
    net["connections"][rng.random(net["connections"].shape) >= 0.6] = 0
    net["weights"][...] *= net["connections"]
    dn = parity.device_from_oracle(snn, net)
    dn.set_history(voltage=True, spikes=True)
    dn.run(70)
    assert dn.stat("persistent_run_launches") == 1
    dn.set_option("run_resident_spin_limit", 1 << 11)
    dn.set_option("run_resident_fault_step", 23)
    dn.run(60)
    assert dn.stat("persistent_run_fallbacks") == 1
    net.run(130, voltage_history=True, spike_history=True)
    parity.assert_state_equal(net, parity.pull_state(dn, net))
    assert np.array_equal(dn.spike_history(3), net.spike_history)
    assert np.array_equal(parity.bits(dn.voltage_history(3)), parity.bits(net.voltage_history))
    dn.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("SNN_RANDOM_SEEDS_PERSISTENT", "32"))))     # (env: a longer campaign)
def test_random_electrical_networks(snn, seed):
    net, calls, history, counts = draw(seed)
    outs = []
    for persistent in (1, 0):
        dn = parity.device_from_oracle(snn, net)
        dn.set_option("persistent_run", persistent)
        dn.set_history(voltage=history, spikes=history)
        dn.set_reduced_history(spike_counts=counts)
        for c in calls:
            dn.run(c)
        out = {"state": parity.pull_state(dn, net), "launches": dn.stat("persistent_run_launches"), "w": dn.get_graph_rows(0, net.n_tot)[0],
               "gave_up": dn.stat("persistent_run_fallbacks")}
        for i, _, _ in net.layout.lattices:
            if history:
                out[("v", i)], out[("s", i)] = dn.voltage_history(i), dn.spike_history(i)
            if counts:
                out[("c", i)] = dn.spike_counts(i)
        for i, _, _ in net.layout.st_lattices:
            if history:
                out[("v", i)] = dn.voltage_history(i)
        dn.close()
        outs.append(out)
    # (a run whose workgroups lost sight of each other -- a device shared with many other processes -- is rolled back and repeated with
    # one launch per step, and the handle stays in that form: fewer one-launch runs then, the same results)
    want = sum(1 for c in calls if c >= 4)
    assert (outs[0]["launches"] == want if not outs[0]["gave_up"] else outs[0]["launches"] <= want) and outs[1]["launches"] == 0
    net.run(sum(calls), voltage_history=history, spike_history=history, spike_counts=counts,
            st_voltage_history=history and net.n_cells > 0)
    rng = net.layout.ranges()
    for out in outs:
        parity.assert_state_equal(net, out["state"])
        assert np.array_equal(parity.bits(out["w"]), parity.bits(np.where(net["connections"] != 0, net["weights"], np.float32(0))))
        for i, _, _ in net.layout.st_lattices:
            first, count, _ = rng[i]
            if history:
                assert np.array_equal(parity.bits(out[("v", i)]), parity.bits(net.st_voltage_history[:, first:first + count]))
        for i, _, _ in net.layout.lattices:
            first, count, _ = rng[i]
            if history:
                assert np.array_equal(out[("s", i)], net.spike_history[:, first:first + count])
                assert np.array_equal(parity.bits(out[("v", i)]), parity.bits(net.voltage_history[:, first:first + count]))
            if counts:
                assert np.array_equal(np.asarray(out[("c", i)]).ravel(), net.spike_counts[first:first + count])



@pytest.mark.parametrize("seed", range(int(os.environ.get("SNN_RANDOM_SEEDS_FAULT", "24"))))          # (env: a longer campaign)
def test_random_fault_injection(snn, seed):
    """The give-up -> rollback -> per-step-repeat path of the one-launch run on RANDOM networks (cells, histories, several run
    calls, voltages rewritten between calls): a random step of a random call withholds workgroup 0's voltages (test hook
    "run_resident_fault_step", spin limit lowered), the handle may be switched back to the one-launch form afterwards and
    faulted again.  Everything snn_run hands back must be what the oracle gives, whatever mixture of forms the calls took.
    A mismatch leaves a repro bundle (tests/repro.py)."""
    import repro
    net, calls, history, counts = draw(seed)
    rng = np.random.default_rng(90_000 + seed)
    calls = calls + [int(rng.integers(8, 90))]
    dn = parity.device_from_oracle(snn, net)
    dn.set_history(voltage=history, spikes=history)
    dn.set_reduced_history(spike_counts=counts)
    dn.set_option("run_resident_spin_limit", 1 << int(rng.integers(8, 13)))
    nn = net.n_neurons
    alone = nn <= 64 and nn + net.n_cells <= 1024           # one workgroup: nothing travels, nothing can be withheld
    lo, hi = {ob.IZHIKEVICH: (-65, 30), ob.LIF: (-80, -50), ob.HH: (-75, -40), ob.LEAKY_IZHIKEVICH: (-65, 30)}.get(net.model, (-75, -56))
    rngs = net.layout.ranges()
    persistent, want_fallbacks, want_launches, log = True, 0, 0, []
    hist = {"v": [], "s": [], "stv": []}
    for k, c in enumerate(calls):
        fault = int(rng.integers(1, c)) if persistent and c >= 4 and rng.random() < 0.6 else 0     # 1 <= fault < c
        dn.set_option("run_resident_fault_step", fault)
        dn.run(c)
        if persistent and c >= 4:
            if fault and not alone:
                want_fallbacks, persistent = want_fallbacks + 1, False
            else:
                want_launches += 1
        net.run(c, voltage_history=history, spike_history=history, spike_counts=counts, st_voltage_history=history and net.n_cells > 0)
        if history:
            hist["v"].append(net.voltage_history), hist["s"].append(net.spike_history)
            if net.n_cells:
                hist["stv"].append(net.st_voltage_history)
        entry = {"call": k, "steps": c, "fault_step": fault, "stats_after": repro.device_stats(dn)}
        if not persistent and rng.random() < 0.5:
            dn.set_option("persistent_run", 1)
            persistent, entry["switched_back_on"] = True, True
        if nn and rng.random() < 0.4:                        # a write behind the stepper's back between two calls
            v = ob.uniform_array(seed * 100 + k, nn, lo, hi)
            net["current_voltage"] = v
            for i, _, _ in net.layout.lattices:
                first, count, _ = rngs[i]
                if count:
                    dn.set_attr(i, "current_voltage", v[first:first + count])
            entry["voltages_rewritten"] = True
        log.append(entry)
    obs, ref = {}, {}
    for name, a in parity.pull_state(dn, net).items():
        obs[f"state/{name}"], ref[f"state/{name}"] = a, net[name]
    for i, _, _ in net.layout.lattices:
        first, count, _ = rngs[i]
        if history:
            obs[f"spikes/{i}"], ref[f"spikes/{i}"] = dn.spike_history(i), np.concatenate(hist["s"])[:, first:first + count]
            obs[f"voltage/{i}"], ref[f"voltage/{i}"] = dn.voltage_history(i), np.concatenate(hist["v"])[:, first:first + count]
        if counts:
            obs[f"counts/{i}"], ref[f"counts/{i}"] = np.asarray(dn.spike_counts(i)).ravel(), net.spike_counts[first:first + count]
    for i, _, _ in net.layout.st_lattices:
        first, count, _ = rngs[i]
        if history:
            obs[f"voltage/{i}"], ref[f"voltage/{i}"] = dn.voltage_history(i), np.concatenate(hist["stv"])[:, first:first + count]
    obs["clock"], ref["clock"] = np.array([dn.clock]), np.array([net.clock])
    obs["weights"], ref["weights"] = dn.get_graph_rows(0, net.n_tot)[0], np.where(net["connections"] != 0, net["weights"], np.float32(0))
    stats = repro.device_stats(dn)
    dn.close()
    diffs = repro.differences(obs, ref)
    quiet = "SNN_CAMPAIGN" not in os.environ                # a shared device may fail the co-residency probe: no form is promised then
    forms_ok = not quiet or (stats["persistent_run_fallbacks"] == want_fallbacks and stats["persistent_run_launches"] == want_launches)
    if diffs or not forms_ok:
        meta = {"test": "test_gpu_persistent_run.py::test_random_fault_injection", "seed": seed, "calls": calls, "log": log,
                "history": history, "counts": counts, "stats": stats, "want_fallbacks": want_fallbacks,
                "want_launches": want_launches, "differences": diffs}
        path = repro.dump(f"fault-{seed}", meta, obs, ref)
        raise AssertionError(f"seed {seed}: {repro.describe(diffs) or 'step forms differ from the plan'} (bundle: {path})")


def build_with_cells(model, lattices, st_lattices, st_kind, seed, density=0.5):
    """electrical-only network with Poisson / Rate spike-train lattices as presynaptic rows (no transmitters, no plasticity)"""
    net = parity.make_oracle(parity.Layout(lattices, st_lattices), model=model, st_kind=st_kind, electrical=True, chemical=False)
    n, nc = net.n_neurons, net.n_cells
    rng = np.random.default_rng(seed)
    lo, hi = {ob.IZHIKEVICH: (-65, 30), ob.LIF: (-80, -50), ob.HH: (-75, -40)}.get(model, (-75, -56))
    net["current_voltage"] = ob.uniform_array(seed, n, lo, hi)
    net["gap_conductance"] = ob.uniform_array(seed + 1, n, 1.0, 10.0)
    if model in (ob.LIF, ob.QIF):
        net["tref"] = ob.uniform_array(seed + 2, n, 0.3, 1.5)
        net["tau_m"] = 10.0
    net["nt_flags"][...] = 0
    net["st_nt_flags"][...] = 0
    net["st_seed"] = rng.integers(1, 2**32 - 1, nc, dtype=np.uint32)
    net["st_chance_of_firing"] = ob.uniform_array(seed + 4, nc, 0.0, 0.08)
    net["st_rate"] = ob.uniform_array(seed + 5, nc, 0.0, 6.0)                # 0: a cell that never fires
    net["st_refractoriness"][...] = rng.integers(0, 2, nc)                   # DeltaDirac / ExponentialDecay per cell
    net["st_k"] = ob.uniform_array(seed + 11, nc, 20.0, 10000.0)
    net.fill_graph(seed + 6, 0.2, 2.0)
    net["connections"][...] &= (rng.random(net["connections"].shape) < density)
    net["weights"][...] *= net["connections"]
    net["do_plasticity"] = 0
    return net


CELL_CASES = [
    (ob.IZHIKEVICH, [(0, 6, 7)], [(5, 3, 4)], ob.ST_POISSON, 61),                      # cells inside the first 64-row block
    (ob.IZHIKEVICH, [(0, 16, 16)], [(5, 16, 16)], ob.ST_POISSON, 62),                  # one cell per neuron (configs[4]'s wiring, small)
    (ob.LIF, [(1, 12, 13), (3, 5, 5)], [(0, 4, 9), (2, 2, 2)], ob.ST_RATE, 63),        # two lattices, two spike-train lattices
    (ob.IZHIKEVICH, [(0, 31, 32)], [(9, 8, 9)], ob.ST_POISSON, 64),                    # cells straddle the first row group's end
    (ob.QIF, [(0, 36, 36)], [(1, 20, 20)], ob.ST_RATE, 65),                            # two row groups, cells in the second
    (ob.IZHIKEVICH, [(0, 45, 45)], [(7, 30, 30)], ob.ST_POISSON, 66),                  # three row groups
]


@pytest.mark.parametrize("model,lattices,st_lattices,st_kind,seed", CELL_CASES)
def test_networks_with_spike_train_cells(snn, model, lattices, st_lattices, st_kind, seed):
    """Spike-train cells are rows every workgroup advances by itself (thread = cell): same results as k_spike_trains."""
    net = build_with_cells(model, lattices, st_lattices, st_kind, seed)
    calls = [120, 2, 61]
    outs = []
    for persistent in (1, 0):
        dn = parity.device_from_oracle(snn, net)
        dn.set_option("persistent_run", persistent)
        dn.set_history(voltage=True, spikes=True)
        for c in calls:
            dn.run(c)
        out = {"state": parity.pull_state(dn, net), "launches": dn.stat("persistent_run_launches"), "w": dn.get_graph_rows(0, net.n_tot)[0]}
        for i, _, _ in list(net.layout.lattices) + list(net.layout.st_lattices):
            out[("v", i)] = dn.voltage_history(i)
        for i, _, _ in net.layout.lattices:
            out[("s", i)] = dn.spike_history(i)
        dn.close()
        outs.append(out)
    assert outs[0]["launches"] == 2 and outs[1]["launches"] == 0
    net.run(sum(calls), voltage_history=True, spike_history=True, st_voltage_history=True)
    assert (net.st_voltage_history > 0).sum() > 10 and net.spike_history.sum() > 0
    rng = net.layout.ranges()
    for out in outs:
        parity.assert_state_equal(net, out["state"])
        assert np.array_equal(parity.bits(out["w"]), parity.bits(np.where(net["connections"] != 0, net["weights"], np.float32(0))))
        for i, _, _ in net.layout.lattices:
            first, count, _ = rng[i]
            assert np.array_equal(out[("s", i)], net.spike_history[:, first:first + count])
            assert np.array_equal(parity.bits(out[("v", i)]), parity.bits(net.voltage_history[:, first:first + count]))
        for i, _, _ in net.layout.st_lattices:
            first, count, _ = rng[i]
            assert np.array_equal(parity.bits(out[("v", i)]), parity.bits(net.st_voltage_history[:, first:first + count]))


@pytest.mark.parametrize("rows,cols,cells,steps", [(32, 32, 0, 40000), (64, 64, 0, 20000), (30, 30, 20, 20000)])
def test_long_runs_agree(snn, rows, cols, cells, steps):
    """Tens of thousands of steps in one launch (the hand-offs between workgroups are timing dependent: a protocol error would
    show as a stall or as a diverging trajectory) against one launch per step: identical final state, device against device."""
    if cells:
        net = build_with_cells(ob.IZHIKEVICH, [(0, rows, cols)], [(5, cells, cells)], ob.ST_POISSON, 71, density=0.2)
    else:
        net = build(ob.IZHIKEVICH, rows, cols, 72, density=0.3)
    states = []
    for persistent in (1, 0):
        dn = parity.device_from_oracle(snn, net)
        dn.set_option("persistent_run", persistent)
        dn.set_reduced_history(spike_counts=True)
        dn.run(steps // 2)
        dn.run(steps - steps // 2)
        states.append((parity.pull_state(dn, net), [dn.spike_counts(i) for i, _, _ in net.layout.lattices],
                       dn.stat("persistent_run_steps")))
        dn.close()
    assert states[0][2] == steps and states[1][2] == 0
    for name in states[0][0]:
        assert np.array_equal(parity.bits(states[0][0][name]), parity.bits(states[1][0][name])), name
    for a, b in zip(states[0][1], states[1][1]):
        assert np.array_equal(a, b) and int(np.asarray(a).sum()) > 0


@pytest.mark.parametrize("rows,cols,cells,model,fault", [(32, 32, 0, ob.IZHIKEVICH, 37), (48, 48, 0, ob.IZHIKEVICH, 5),
                                                         (20, 20, 100, ob.IZHIKEVICH, 11), (24, 24, 0, ob.HH, 2)])
def test_a_run_that_gives_up_is_rolled_back_and_repeated_per_step(snn, rows, cols, cells, model, fault):
    """include/snn_amd.h, failure semantics of "persistent_run": a workgroup withholds its voltages after step `fault` of
    the one launch (test hook), every reader of its columns times out (spin limit lowered so that this takes milliseconds),
    and snn_run must hand back exactly what a per-step run gives -- state, clocks, histories, spike totals, generator
    state of the cells -- in the SAME call, with the handle left in per-step mode and usable."""
    if cells:
        lay = parity.Layout([(3, rows, cols)], [(1, 10, cells // 10)])
        net = parity.make_oracle(lay, model=model, st_kind=ob.ST_POISSON)
        net["st_chance_of_firing"] = 0.05
        n = net.n_neurons
        net["current_voltage"] = ob.uniform_array(90 + fault, n, -65, 30)
        net["gap_conductance"] = ob.uniform_array(95, n, 2.0, 10.0)
        net.fill_graph(91, 0.5, 1.5)
        rng = np.random.default_rng(92)
        net["connections"][rng.random(net["connections"].shape) >= 0.75] = 0
        net["weights"][...] *= net["connections"]
    else:
        net = build(model, rows, cols, 90 + fault)
    dn = parity.device_from_oracle(snn, net)
    dn.set_history(voltage=True, spikes=True)
    dn.set_reduced_history(spike_counts=True)
    dn.run(60)                                          # a clean one-launch run first: granule slots and tags are in use
    assert dn.stat("persistent_run_launches") == 1 and dn.stat("persistent_run_fallbacks") == 0
    dn.set_option("run_resident_spin_limit", 1 << 12)
    dn.set_option("run_resident_fault_step", fault)
    dn.run(80)                                          # gives up at step `fault`, rolled back, repeated per step
    assert dn.stat("persistent_run_fallbacks") == 1
    assert dn.stat("persistent_run_launches") == 1 and dn.stat("persistent_run_steps") == 60
    assert dn.clock == 140
    dn.set_option("run_resident_fault_step", 0)
    dn.run(50)                                          # the handle keeps one launch per step ...
    assert dn.stat("persistent_run_launches") == 1
    dn.set_option("run_resident_spin_limit", 0)         # (0 = the default limit)
    dn.set_option("persistent_run", 1)                  # ... until it is switched back on
    dn.run(40)
    assert dn.stat("persistent_run_launches") == 2 and dn.stat("persistent_run_fallbacks") == 1
    net.run(230, voltage_history=True, spike_history=True, spike_counts=True)
    parity.assert_state_equal(net, parity.pull_state(dn, net))
    assert np.array_equal(dn.spike_history(3), net.spike_history)
    assert np.array_equal(parity.bits(dn.voltage_history(3)), parity.bits(net.voltage_history))
    assert np.array_equal(dn.spike_counts(3), net.spike_counts)
    assert net.spike_history.sum() > 0 or model != ob.IZHIKEVICH
    dn.close()


def test_a_callers_stream_keeps_one_launch_per_step_and_says_so(snn):
    """snn_set_stream: the one-launch run reads its outcome after a host synchronisation, so a handle on a caller's stream keeps
    the per-step form -- counted by the statistic "persistent_run_external_stream", gone when the stream is handed back"""
    import torch
    net = build(ob.IZHIKEVICH, 12, 12, 77)
    dn = parity.device_from_oracle(snn, net)
    side = torch.cuda.Stream()
    dn.set_stream(side.cuda_stream)
    dn.run(40)
    dn.synchronize()
    assert dn.stat("persistent_run_launches") == 0 and dn.stat("persistent_run_external_stream") == 1
    dn.set_stream(None)
    dn.run(40)
    assert dn.stat("persistent_run_launches") == 1 and dn.stat("persistent_run_external_stream") == 1
    net.run(80)
    parity.assert_state_equal(net, parity.pull_state(dn, net))
    dn.close()
