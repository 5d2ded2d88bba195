"""Randomised differential test: many small networks drawn from a seeded generator -- model (9), neurotransmitter /
receptor kinetics (4 x 3), synapse kinds, lattice shapes, connectivity density, spike-train inputs (Poisson, rate,
preset, BCM Poisson), per-lattice plasticity (none / STDP / BCM / reward-modulated with a random reward sequence), graph form (dense /
sparse), whole or sharded stepping, run split into several calls, reduced histories with a random stride -- each
compared bit for bit with the oracle."""
import os

import numpy as np
import pytest

import checkpoint
import oracle_binding as ob
import parity
import repro

pytestmark = pytest.mark.gpu

MODELS = [ob.IZHIKEVICH, ob.LIF, ob.HH, ob.QIF, ob.SIMPLE_LIF, ob.ADAPTIVE_LIF, ob.ADAPTIVE_EXP_LIF, ob.LEAKY_IZHIKEVICH,
          ob.BCM_IZHIKEVICH]


def draw(seed):
    rng = np.random.default_rng(seed)
    model = MODELS[seed % len(MODELS)]
    n_lat = int(rng.integers(1, 4))
    lattices = [(int(i * 2 + rng.integers(0, 2)), int(rng.integers(1, 9)), int(rng.integers(1, 12))) for i in range(n_lat)]
    st_kind = [ob.ST_NONE, ob.ST_POISSON, ob.ST_RATE, ob.ST_PRESET, ob.ST_BCM_POISSON][int(rng.integers(0, 5))]
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
              ob.LEAKY_IZHIKEVICH: (-65, 30), ob.BCM_IZHIKEVICH: (-65, 30)}[model]
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
        net["st_refractoriness"][...] = rng.integers(0, 2, nc)        # DeltaDirac / ExponentialDecay per cell
        net["st_k"] = ob.uniform_array(seed + 11, nc, 20.0, 10000.0)
        if st_kind == ob.ST_PRESET:
            net.set_firing_times([list(rng.uniform(0.3, 5.0, int(k))) for k in rng.integers(0, 4, nc)])
    net.fill_graph(seed + 6, -0.5, 2.0, with_diagonal=bool(rng.integers(0, 2)))
    density = float(rng.choice([0.05, 0.3, 0.8, 1.0]))
    net["connections"][...] &= (rng.random(net["connections"].shape) < density)
    net["weights"][...] *= net["connections"]
    for slot in range(len(lattices)):
        mode = int(rng.integers(0, 3))                  # 0 static weights, 1 STDP, 2 reward-modulated (R-STDP traces)
        net["do_plasticity"][slot] = int(mode == 1)
        if mode == 1 and model == ob.BCM_IZHIKEVICH and rng.integers(0, 2):
            net["plasticity_kind"][slot] = 1                    # the BCM rule instead of STDP
            net["bcm_decay"][slot] = float(rng.uniform(0.01, 0.2))
            net["bcm_average_scalar"][slot] = float(rng.uniform(0.1, 1.0))
        net["stdp_a_plus"][slot] = float(rng.uniform(0.5, 2.5))
        net["stdp_tau_minus"][slot] = float(rng.uniform(2.0, 6.0))
        if mode == 2:
            net["rm_do_modulation"][slot] = 1
            net["rm_dopamine"][slot] = float(rng.uniform(0.0, 0.02))
            net["rm_tau_d"][slot] = float(rng.uniform(2.0, 10.0))
            net["rm_tau_c"][slot] = float(rng.uniform(0.02, 0.2))
            net["rm_a_plus"][slot] = float(rng.uniform(0.001, 0.01))
            net["rm_a_minus"][slot] = float(rng.uniform(0.001, 0.01))
    net["traces"][...] = ob.uniform_array(seed + 9, net["traces"].size, -0.001, 0.001).reshape(net["traces"].shape)
    net["traces"][...] *= net["connections"]
    dt = 0.01 if model == ob.HH else float(rng.choice([0.05, 0.1, 0.2]))
    net["dt"] = dt
    net["st_dt"] = dt
    net["stdp_dt"] = dt
    net["rm_dt"] = dt
    net["bcm_dt"] = dt
    if model == ob.BCM_IZHIKEVICH:
        net["bcm_window"] = ob.uniform_array(seed + 12, nn, 5 * dt, 40 * dt)
        net["bcm_period"] = rng.integers(1, 6, nn).astype(np.uint32)
    if st_kind == ob.ST_BCM_POISSON:
        net["st_bcm_window"] = ob.uniform_array(seed + 13, nc, 5 * dt, 40 * dt)
        net["st_bcm_period"] = rng.integers(1, 6, nc).astype(np.uint32)
    plan = dict(csr=bool(rng.integers(0, 2)), shards=int(rng.choice([1, 1, 2, 3])), by_lattice=bool(rng.integers(0, 2)),
                steps=int(rng.integers(80, 260)), calls=int(rng.integers(1, 4)), stride=int(rng.choice([1, 1, 2, 5])))
    if net["plasticity_kind"].any():
        plan["shards"] = 1                                      # BCM activities are not exchanged between shards
    plan["rewards"] = None
    if net["rm_do_modulation"].any():
        r = ob.uniform_array(seed + 10, plan["steps"], -0.02, 0.03)
        r[::4] = 0.0
        plan["rewards"] = r
    return net, plan


SWITCHES = {"fused_step": (0, 1), "defer_stdp": (0, 1, 2, 3), "defer_rstdp": (0, 1), "uniform_params": (0, 1), "persistent_run": (0, 1), "persistent_chem": (0, 1),
            "cells_in_step": (0, 1), "csr_xcd_bands": (0, 1), "update_packs": (0, 1), "input_shape": (0, 1, 2), "stdp_columns_form": (0, 1), "stdp_small": (0, 1)}


def tuning_switches(seed):
    """results never depend on the tuning switches (include/snn_amd.h): two seeds in three run with a random setting of them"""
    if seed % 3 == 0:
        return {}
    rng = np.random.default_rng(50_000 + seed)
    return {name: int(rng.choice(values)) for name, values in SWITCHES.items() if rng.integers(0, 2)}


RM_KEYS = ("rm_dopamine", "rm_tau_d", "rm_tau_c", "rm_a_plus", "rm_a_minus", "rm_tau_plus", "rm_tau_minus", "rm_dt")


def csr_order(net, dense, posts):
    """values of a dense [n_tot][n_neurons] array in the CSR-by-post edge order of the columns `posts`"""
    ptr, pre, _ = parity.csr_for_posts(net, posts)
    return (np.concatenate([dense[pre[ptr[k]:ptr[k + 1]], q] for k, q in enumerate(posts)]) if len(posts)
            else np.zeros(0, np.float32))


def make_handle(snn, net, plan, shard=None):
    dn = parity.device_from_oracle(snn, net, shard=shard, csr=plan["csr"],
                                   by_lattice=bool(shard is not None and plan["csr"] and plan.get("by_lattice")))
    if plan["rewards"] is not None:
        for slot, (i, _, _) in enumerate(net.layout.lattices):
            if net["rm_do_modulation"][slot]:
                dn.set_reward_modulator(i, *(float(net[k][slot]) for k in RM_KEYS), do_modulation=True)
        if plan["csr"]:
            dn.set_traces_csr(csr_order(net, net["traces"], dn.owned))
        elif net.n_neurons and net.n_tot:
            dn.set_trace_rows(0, net["traces"])
    return dn


def check_modulation(dn, net, plan):
    if plan["rewards"] is None:
        return
    b, e = dn.post_begin, dn.post_end
    if plan["csr"]:
        assert np.array_equal(parity.bits(dn.get_traces_csr()), parity.bits(csr_order(net, net["traces"], dn.owned)))
    elif net.n_neurons and net.n_tot:
        t = dn.get_trace_rows(0, net.n_tot)
        assert np.array_equal(parity.bits(t[:, b:e]), parity.bits(net["traces"][:, b:e]))
    for slot, (i, _, _) in enumerate(net.layout.lattices):
        if net["rm_do_modulation"][slot]:
            assert np.array_equal(parity.bits(np.array([dn.dopamine(i)])), parity.bits(net["rm_dopamine"][slot:slot + 1]))


def tracked(seed):
    """two seeds in three compare device and oracle at EVERY run-call boundary (checkpoint.Tracker: a mismatch is localised to one
    call and that call executed again from a checkpoint); the comparison is a getter and flushes what a run call leaves pending, so
    the third keeps the calls back to back"""
    return seed % 3 != 1


def device_unsharded(snn, seed, track=False):
    """the device side of an unsharded case: every compared array under the key the oracle side uses + snn_get_stat counters"""
    net, plan = draw(1000 + seed)
    steps = plan["steps"]
    dn = make_handle(snn, net, plan)
    tr = checkpoint.Tracker(snn, dn, net, plan, f"random-{seed}", enabled=track)      # (`net`: a second oracle, stepped call by call)
    for name, value in tuning_switches(seed).items():
        tr.set_option(name, value)
    dn.set_history(voltage=True, spikes=True)
    dn.set_reduced_history(True, True, True)
    dn.set_history_stride(plan["stride"])
    if plan["rewards"] is not None:
        for r in plan["rewards"]:
            dn.run_with_reward(float(r))
    else:
        done = 0
        for c in range(plan["calls"]):
            k = steps // plan["calls"] if c < plan["calls"] - 1 else steps - done
            if tr.enabled:
                tr.run(k)
            else:
                dn.run(k)
            done += k
    obs = {}
    for slot, (i, _, _) in enumerate(net.layout.lattices):
        obs[f"spikes/{i}"], obs[f"voltage/{i}"], obs[f"counts/{i}"] = dn.spike_history(i), dn.voltage_history(i), dn.spike_counts(i)
        if net.layout.ranges()[i][1]:
            obs[f"average_voltage/{i}"], obs[f"eeg/{i}"] = dn.average_voltage_history(i), dn.eeg_history(i)
    for name, a in parity.pull_state(dn, net).items():
        obs[f"state/{name}"] = a
    if net.n_tot and net.n_neurons:
        if plan["csr"]:
            obs["graph/csr_weights"] = dn.get_graph_csr()
        else:
            obs["graph/weights"], obs["graph/connections"] = dn.get_graph_rows(0, net.n_tot)
    if plan["rewards"] is not None:
        if plan["csr"]:
            obs["traces"] = dn.get_traces_csr()
        elif net.n_neurons and net.n_tot:
            obs["traces"] = dn.get_trace_rows(0, net.n_tot)
        for slot, (i, _, _) in enumerate(net.layout.lattices):
            if net["rm_do_modulation"][slot]:
                obs[f"dopamine/{i}"] = np.array([dn.dopamine(i)])
    obs["clock"] = np.array([dn.clock], np.int64)
    stats = repro.device_stats(dn)
    dn.close()
    return obs, stats


def oracle_unsharded(seed, state_names=()):
    """the oracle side under the same keys (state arrays for the names the device downloaded)"""
    net, plan = draw(1000 + seed)
    steps = plan["steps"]
    net.run(steps, voltage_history=True, spike_history=True, summaries=True, spike_counts=True, rewards=plan["rewards"])
    keep = np.arange(0, steps, plan["stride"])
    rng = net.layout.ranges()
    ref = {}
    for slot, (i, _, _) in enumerate(net.layout.lattices):
        first, count, _ = rng[i]
        ref[f"spikes/{i}"] = net.spike_history[keep, first:first + count]
        ref[f"voltage/{i}"] = net.voltage_history[keep, first:first + count]
        ref[f"counts/{i}"] = net.spike_counts[first:first + count]
        if count:
            ref[f"average_voltage/{i}"], ref[f"eeg/{i}"] = net.avg_history[keep, slot], net.eeg_history[keep, slot]
    for name in state_names:
        ref[f"state/{name}"] = net[name]
    if net.n_tot and net.n_neurons:
        if plan["csr"]:
            ref["graph/csr_weights"] = parity.csr_for_posts(net, np.arange(net.n_neurons))[2]
        else:
            oc = net["connections"].astype(np.uint32)
            ref["graph/weights"], ref["graph/connections"] = np.where(oc != 0, net["weights"], np.float32(0)), oc
    if plan["rewards"] is not None:
        if plan["csr"]:
            ref["traces"] = csr_order(net, net["traces"], np.arange(net.n_neurons))
        elif net.n_neurons and net.n_tot:
            ref["traces"] = net["traces"]
        for slot, (i, _, _) in enumerate(net.layout.lattices):
            if net["rm_do_modulation"][slot]:
                ref[f"dopamine/{i}"] = net["rm_dopamine"][slot:slot + 1]
    ref["clock"] = np.array([net.clock], np.int64)
    return ref


def run_unsharded(snn, seed):
    """One execution of an unsharded case.  Returns None, or -- after leaving a repro bundle -- the description of the
    mismatch.  A mismatch is followed by one more oracle run and two more device runs in this process (repro.triage)."""
    obs, stats = device_unsharded(snn, seed, track=tracked(seed))
    names = [k[len("state/"):] for k in obs if k.startswith("state/")]
    ref = oracle_unsharded(seed, names)
    diffs = repro.differences(obs, ref)
    if not diffs:
        return None
    _, plan = draw(1000 + seed)
    meta = {"test": "test_gpu_randomized.py::test_random_network", "seed": seed, "switches": tuning_switches(seed),
            "plan": {k: v for k, v in plan.items() if k != "rewards"}, "rewards": plan["rewards"] is not None,
            "stats": stats, "differences": diffs, "environment": {k: v for k, v in os.environ.items() if k.startswith(("SNN_", "AMD_", "HIP_", "HSA_", "OMP_"))}}
    meta["triage"] = repro.triage(lambda: device_unsharded(snn, seed), lambda: oracle_unsharded(seed, names), ref)
    path = repro.dump(f"random-{seed}", meta, obs, ref)
    return f"seed {seed}: {repro.describe(diffs)} (bundle: {path})"


# SNN_RANDOM_SEEDS=n widens the sweep (an occasional long campaign; the suite keeps 72)
# (583: found by a 900-seed campaign -- a range-set sparse shard stepped rows of its blocks that it does not own)
def threaded(seed):
    """cases whose shard handles are stepped by the library's own loop, one host thread per rank (run_sharded below)"""
    _, plan = draw(1000 + seed)
    return plan["shards"] > 1 and plan["rewards"] is None and seed % 2 == 1


@pytest.mark.parametrize("seed", [pytest.param(s, marks=pytest.mark.emulated_ranks) if threaded(s) else s
                                  for s in sorted(set(range(int(os.environ.get("SNN_RANDOM_SEEDS", "72")))) | {583})])
def test_random_network(snn, seed):
    net, plan = draw(1000 + seed)
    if plan["shards"] == 1:
        problem = run_unsharded(snn, seed)
        assert problem is None, problem
        return
    try:
        run_sharded(snn, seed, net, plan)
    except AssertionError as e:
        meta = {"test": "test_gpu_randomized.py::test_random_network (sharded)", "seed": seed, "switches": tuning_switches(seed),
                "plan": {k: v for k, v in plan.items() if k != "rewards"}, "assertion": str(e)[:4000]}
        path = repro.dump(f"random-sharded-{seed}", meta)
        raise AssertionError(f"seed {seed} (bundle: {path}): {e}") from e


def run_sharded(snn, seed, net, plan):
    import torch
    from snn_amd import parallel
    steps = plan["steps"]
    switches = tuning_switches(seed)
    g = plan["shards"]
    handles = [make_handle(snn, net, plan, shard=(r, g)) for r in range(g)]
    for h in handles:
        for name, value in switches.items():
            h.set_option(name, value)
        h.set_history(voltage=True, spikes=True)              # a shard records its own neurons (global 64-blocks of the raster)
        h.set_reduced_history(False, False, True)
        h.set_history_stride(plan["stride"])
    if plan["rewards"] is None and seed % 2:
        # the library's own loop (snn_run_sharded), one host thread per rank, the collectives replaced by device copies
        from test_gpu_library_loop_threads import run_ranks
        tc = parallel.ThreadCollectives(g, torch.device("cuda", 0))
        try:
            first = steps // plan["calls"]
            run_ranks(handles, tc, [first, steps - first] if plan["calls"] > 1 else [steps])
        finally:
            tc.close()
    else:
        ex = parallel.LocalExchange(handles, torch.device("cuda", 0), halo=plan["csr"] and bool(plan.get("halo", True)))
        for step in range(steps):
            for h in handles:
                if plan["rewards"] is not None:
                    h.apply_reward(float(plan["rewards"][step]))
                h.step_begin_local()
                h.step_begin()
            ex.exchange()
            for h in handles:
                h.step_end()
    net.run(steps, voltage_history=True, spike_history=True, spike_counts=True, rewards=plan["rewards"])
    keep = np.arange(0, steps, plan["stride"])
    rngs = net.layout.ranges()
    for h in handles:
        check_modulation(h, net, plan)
        st = parity.pull_state(h, net)
        parity.assert_shard_view_equal(h, st, net)
        own = np.zeros(net.n_neurons, bool)
        own[h.owned] = True
        for i, _, _ in net.layout.lattices:
            first, count, _ = rngs[i]
            m = own[first:first + count]
            if not m.any():
                continue
            assert np.array_equal(h.spike_history(i)[:, m], net.spike_history[keep, first:first + count][:, m])
            assert np.array_equal(parity.bits(h.voltage_history(i)[:, m]), parity.bits(net.voltage_history[keep, first:first + count][:, m]))
            assert np.array_equal(h.spike_counts(i)[m], net.spike_counts[first:first + count][m])
        b, e = h.post_begin, h.post_end
        for name in ("rc_r", "rc_current"):
            assert np.array_equal(parity.bits(st[name][h.owned]), parity.bits(net[name][h.owned])), name
        if h.csr:
            parity.assert_graph_equal(net, h)
        elif net.n_neurons and net.n_tot:
            w, c = h.get_graph_rows(0, net.n_tot)
            ow = np.where(net["connections"] != 0, net["weights"], np.float32(0))
            assert np.array_equal(parity.bits(w[:, b:e]), parity.bits(ow[:, b:e]))
        h.close()
