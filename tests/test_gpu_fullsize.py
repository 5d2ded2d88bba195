"""Parity at BASELINE.json's full sizes.

* configs[1] (256x256 Izhikevich, dense, 17.18 GB of weights generated on the device): ALL 65 536 postsynaptic
  neurons, teacher-forced -- for each step the oracle recomputes, from the GPU's own state S(t) and the same
  counter-based weights (a 21.5 GB host copy, or column windows that cover the population when memory is short),
  the inputs and the update of every neuron and must reproduce the GPU's S(t+1) bit for bit.
* configs[2] (128x128 Hodgkin-Huxley + Na/K + Destexhe AMPA, electrical + chemical): full oracle run.
* configs[3]-shaped excitatory/inhibitory network with STDP at a size the host can hold (20 480 neurons).
* configs[3] at full size (81 920 neurons, 26.8 GB): the same for every neuron, plus the deferred STDP of every spiking
  neuron applied to the oracle's host copy and ALL 6.7 G weights compared after the last step.
* configs[4] (4 x 512^2 Izhikevich neurons + 4 x 512^2 Poisson cells, sparse): the oracle's dense matrix would need
  4.4 TB; the oracle's CSR-by-post routine (snn_o_run_csr, same canonical chunk flush) runs the full size next to the
  device, with a vectorised numpy restatement of the sparse step (float32, one rounding per operation) as the second
  witness.
"""
import numpy as np
import pytest

import oracle_binding as ob
import parity

pytestmark = pytest.mark.gpu


def column_windows(n, n_rows, budget_bytes, block=1024):
    """[c0, c1) windows (multiples of `block`) that together cover all n postsynaptic columns, each small enough for a
    host copy of its [n_rows][c1 - c0] weights + connection flags (5 B per synapse) to fit the budget"""
    fit = max(block, int(budget_bytes // (5 * n_rows)) // block * block)
    return [(c0, min(n, c0 + fit)) for c0 in range(0, n, fit)]


def window_net(n, n_lattices, c0, c1, seed, threads):
    """an oracle net of n Izhikevich neurons holding the postsynaptic columns [c0, c1) of the synthetic graph"""
    net = ob.Net(n, model=ob.IZHIKEVICH, n_lattices=n_lattices, dense=False)
    net.arr["weights"] = np.empty((n, c1 - c0), np.float32)
    net.arr["connections"] = np.empty((n, c1 - c0), np.uint8)
    net.w_col0, net.w_ld = c0, c1 - c0
    ob.lib().snn_o_fill_graph_window_blocked(net["weights"].ctypes.data_as(ob.f32p), net["connections"].ctypes.data_as(ob.u8p),
                                             n, n, c0, c1 - c0, 1024, seed, 0.5, 1.5, 0, threads)
    net["gap_conductance"] = 10.0
    net.n_threads = threads
    return net


def test_c2_256x256_all_columns_teacher_forced(snn):
    """BASELINE configs[1] at full size, EVERY postsynaptic neuron: for each step the oracle recomputes, from the
    GPU's own state S(t) and the same counter-based weights, the input sum (all 65 536 presynaptic terms) and the update
    of all 65 536 neurons and must reproduce the GPU's S(t+1) bit for bit.  The host copy of the matrix (21.5 GB) is
    held whole when MemAvailable allows, else in column windows that together cover the population."""
    from snn_amd import synthetic
    rows = cols = 256
    n = rows * cols
    steps = 4
    dn = snn.DeviceNetwork(model=snn.IZHIKEVICH)
    dn.add_lattice(0, rows, cols)
    dn.finalize()
    v0 = synthetic.uniform(1, n, -65.0, 30.0)
    v0[[100, 1000, 32768, n - 5, 5000]] = 40.0          # above threshold: these spike (and reset) in step 0
    dn.set_attr(0, "gap_conductance", np.full(n, 10.0, np.float32))
    dn.set_attr(0, "current_voltage", v0)
    dn.fill_graph_synthetic(2, 0.5, 1.5, with_diagonal=False)

    # the device-side generator against the oracle's, on sampled rows
    for p in (0, 255, 256, 40000, n - 1):
        w, c = dn.get_graph_rows(p, 1)
        want = ob.uniform_array(2, n, 0.5, 1.5, offset=p * n)
        want[p] = 0.0
        assert np.array_equal(w[0].view(np.uint32), want.view(np.uint32)) and c[0].sum() == n - 1 and c[0, p] == 0

    def snapshot():
        return {"current_voltage": dn.get_attr(0, "current_voltage"), "w_value": dn.get_attr(0, "w_value"),
                "is_spiking": dn.get_attr(0, "is_spiking", dtype=np.uint32),
                "last_firing_time": dn.get_attr(0, "last_firing_time", dtype=np.int32)}

    states = [snapshot()]
    for _ in range(steps):
        dn.run(1)
        states.append(snapshot())
    dn.close()
    assert np.isfinite(states[-1]["current_voltage"]).all()
    assert sum(int(s["is_spiking"].sum()) for s in states[1:]) >= 5

    threads = ob.usable_cpus()
    windows = column_windows(n, n, ob.mem_available_bytes() // 3)
    checked = np.zeros(n, bool)
    for c0, c1 in windows:
        net = window_net(n, 1, c0, c1, 2, threads)
        for step in range(steps):
            prev, new = states[step], states[step + 1]
            net["current_voltage"] = prev["current_voltage"]
            net["w_value"] = prev["w_value"]
            net["last_firing_time"] = prev["last_firing_time"]
            net.clock = step
            net.inputs_tiled(c0, c1)
            net.update_neurons(c0, c1)
            where = f"step {step}, columns [{c0}, {c1}) of {len(windows)} window(s)"
            for k in ("current_voltage", "w_value"):
                assert np.array_equal(net[k][c0:c1].view(np.uint32), new[k][c0:c1].view(np.uint32)), (k, where)
            assert np.array_equal(net["is_spiking"][c0:c1], new["is_spiking"][c0:c1]), where
            assert np.array_equal(net["last_firing_time"][c0:c1], new["last_firing_time"][c0:c1]), where
        checked[c0:c1] = True
        del net
    assert checked.all(), f"{int(checked.sum())} of {n} postsynaptic neurons checked"


def test_c3_128x128_hodgkin_huxley_ampa_full_oracle(snn):
    """BASELINE configs[2]: HH defaults dt = 0.01, Destexhe NT + Destexhe receptor (AMPA g = 1, e = 0),
    electrical + chemical, all-to-all, V0 ~ U[-70, -60] seed 3; 40 steps, everything bit-identical."""
    lay = parity.Layout([(0, 128, 128)])
    net = parity.make_oracle(lay, model=ob.HH, nt_kind=ob.NT_DESTEXHE, rc_kind=ob.RC_DESTEXHE, chemical=True)
    n = net.n_neurons
    net["current_voltage"] = ob.uniform_array(3, n, -70.0, -60.0)
    net["nt_flags"][:, 0] = 1
    net["rc_flags"][:, 0] = 1
    net.fill_graph(4, 0.5, 1.5)
    net.n_threads = 16
    dn = parity.device_from_oracle(snn, net)
    dn.set_history(voltage=True, spikes=True)
    dn.run(40)
    net.run(40, voltage_history=True, spike_history=True)
    assert np.array_equal(dn.spike_history(0), net.spike_history)
    assert np.array_equal(parity.bits(dn.voltage_history(0)), parity.bits(net.voltage_history))
    parity.assert_state_equal(net, parity.pull_state(dn, net))
    # north_star tolerance for membrane traces (1e-5 relative) is met with margin zero: they are identical
    dn.close()


def test_c4_shaped_network_with_stdp_20480_neurons(snn):
    """configs[3] scaled to host memory: exc 128x128 (id 1) + inh 64x64 (id 0), dense, STDP on both."""
    lay = parity.Layout([(0, 64, 64), (1, 128, 128)])
    net = parity.make_oracle(lay)
    n = net.n_neurons
    r = lay.ranges()
    inh = slice(r[0][0], r[0][0] + r[0][1])
    net["current_voltage"] = ob.uniform_array(4, n, -65.0, 30.0)
    net["gap_conductance"] = 10.0
    net.fill_graph(5, 0.5, 1.5)
    net["weights"][inh, :] *= -1.0
    net["do_plasticity"] = 1
    net.n_threads = 16
    dn = parity.device_from_oracle(snn, net)
    dn.set_history(voltage=False, spikes=True)
    dn.run(12)
    net.run(12, spike_history=True)
    total = int(net.spike_history.sum())
    assert total > 20, total
    for i in (0, 1):
        first, count, _ = r[i]
        assert np.array_equal(dn.spike_history(i), net.spike_history[:, first:first + count])
    parity.assert_state_equal(net, parity.pull_state(dn, net))
    parity.assert_graph_equal(net, dn)
    dn.close()


def test_c4_full_size_network_with_stdp_all_columns(snn):
    """BASELINE configs[3] at FULL size (256x256 excitatory + 128x128 inhibitory = 81 920 neurons, 26.8 GB of weights
    generated on the device, > 2^32 matrix elements), EVERY postsynaptic neuron and EVERY synapse: per step the oracle
    recomputes inputs and updates of all neurons from the GPU's S(t) (teacher-forced, as in the configs[1] test) and
    applies the deferred STDP of every neuron that spiked to its own host copy of the matrix (33.5 GB, or column windows
    that cover it); after the last step all 6.7 G weights on the device equal the oracle's bit for bit."""
    from snn_amd import synthetic
    n_inh, n_exc = 128 * 128, 256 * 256
    n = n_inh + n_exc
    steps = 3
    dn = snn.DeviceNetwork(model=snn.IZHIKEVICH)
    dn.add_lattice(0, 128, 128)
    dn.add_lattice(1, 256, 256)
    dn.finalize()
    v0 = synthetic.uniform(4, n, -65.0, 30.0)
    hot = np.array([5, 1000, n_inh - 1, n_inh, n_inh + 7, 40000, n - 2])
    v0[hot] = 40.0                                  # spike (and trigger STDP) in step 0
    dn.set_attr(0, "gap_conductance", np.full(n_inh, 10.0, np.float32))
    dn.set_attr(1, "gap_conductance", np.full(n_exc, 10.0, np.float32))
    dn.set_attr(0, "current_voltage", v0[:n_inh])
    dn.set_attr(1, "current_voltage", v0[n_inh:])
    dn.fill_graph_synthetic(5, 0.5, 1.5, with_diagonal=False)
    dn.set_plasticity(0, a_plus=1.5)
    dn.set_plasticity(1, tau_minus=3.0)

    def state(name, dtype=np.float32):
        return np.concatenate([dn.get_attr(0, name, dtype=dtype), dn.get_attr(1, name, dtype=dtype)])

    def snapshot():
        return {"current_voltage": state("current_voltage"), "w_value": state("w_value"),
                "is_spiking": state("is_spiking", np.uint32), "last_firing_time": state("last_firing_time", np.int32)}

    # The synthetic lattice is quiescent by itself (the gap junctions pull everything towards the mean), and STDP only
    # moves a weight between two neurons that have BOTH fired, at different times: further groups are raised above
    # threshold before steps 1 and 2, so that potentiation (t_pre < t_post) and depression occur on known edges.
    inject = {1: np.array([17, n_inh + 100, 60000, 81000]), 2: np.array([5, 2222, n_inh + 7, 70000])}
    before, after = [], []
    for step in range(steps):
        if step in inject:
            v = state("current_voltage")
            v[inject[step]] = 40.0
            dn.set_attr(0, "current_voltage", v[:n_inh])
            dn.set_attr(1, "current_voltage", v[n_inh:])
        before.append(snapshot())
        dn.run(1)
        after.append(snapshot())
    for step, group in ((0, hot), (1, inject[1]), (2, inject[2])):
        assert after[step]["is_spiking"][group].all(), step

    threads = ob.usable_cpus()
    windows = column_windows(n, n, ob.mem_available_bytes() // 3)
    checked = np.zeros(n, bool)
    touched = 0
    for c0, c1 in windows:
        net = window_net(n, 2, c0, c1, 5, threads)
        net["lattice"][n_inh:] = 1
        net["do_plasticity"] = 1
        net["stdp_a_plus"][0] = 1.5
        net["stdp_tau_minus"][1] = 3.0
        for step in range(steps):
            prev, new = before[step], after[step]
            for k in ("current_voltage", "w_value", "last_firing_time"):
                net[k] = prev[k]
            net.clock = step
            net.inputs_tiled(c0, c1)                    # from the oracle's OWN weights: STDP of the earlier steps included
            net.update_neurons(c0, c1)
            where = f"step {step}, columns [{c0}, {c1}) of {len(windows)} window(s)"
            for k in ("current_voltage", "w_value"):
                assert np.array_equal(net[k][c0:c1].view(np.uint32), new[k][c0:c1].view(np.uint32)), (k, where)
            assert np.array_equal(net["is_spiking"][c0:c1], new["is_spiking"][c0:c1]), where
            assert np.array_equal(net["last_firing_time"][c0:c1], new["last_firing_time"][c0:c1]), where
            # plasticity needs the spikes and firing times of ALL neurons (rows): the window's were just checked, the
            # other windows check theirs
            net["is_spiking"] = new["is_spiking"]
            net["last_firing_time"] = new["last_firing_time"]
            net.plasticity(c0, c1)                      # neuron/mod.rs:2308-2417 on this window's columns
        # every weight of the window after the last step
        block = 2048
        for p0 in range(0, n, block):
            nb = min(block, n - p0)
            w, c = dn.get_graph_rows(p0, nb)
            assert np.array_equal(c[:, c0:c1] != 0, net["connections"][p0:p0 + nb] != 0), (p0, c0)
            got, want = w[:, c0:c1], net["weights"][p0:p0 + nb]
            if not np.array_equal(got.view(np.uint32), want.view(np.uint32)):
                bad = np.argwhere(got.view(np.uint32) != want.view(np.uint32))
                r, q = bad[0]
                raise AssertionError(f"{len(bad)} weights differ in rows [{p0}, {p0 + nb}) x columns [{c0}, {c1}); first "
                                     f"({p0 + r}, {c0 + q}): oracle {want[r, q]!r} device {got[r, q]!r}")
            del w, c
        # the updates really happened: an edge from a neuron that fired in step 0 to one that fired in the last step has
        # gained a_plus * exp(-|dt * (t_pre - t_post)| / tau_plus) over the generated weight
        for pp in hot:
            for q in inject[2]:
                if pp != q and c0 <= q < c1:
                    fresh = np.float32(ob.uniform(5, int(pp) * n + int(q), 0.5, 1.5))
                    prm = (2.0, 2.0, 4.5, 3.0, 0.1) if q >= n_inh else (1.5, 2.0, 4.5, 4.5, 0.1)
                    gain = np.float32(ob.lib().snn_o_stdp_delta(0, 2, *prm))
                    if pp not in inject[2]:             # those fire in steps 0 AND 2: their pairs have t_pre == t_post
                        assert net["weights"][pp, q - c0] == np.float32(fresh + gain), (pp, q)
                        touched += 1
        checked[c0:c1] = True
        del net
    assert checked.all()
    assert touched == 5 * 4, touched
    dn.close()


def c5_handle(snn, side, v0, shard=None, by_lattice=False):
    """BASELINE configs[4] on a (shard) handle: 4 Izhikevich lattices + 4 Poisson lattices, CSR rows of the handle"""
    from snn_amd import synthetic
    f32 = np.float32
    m = side * side
    dn = snn.DeviceNetwork(model=snn.IZHIKEVICH, spike_train=snn.ST_POISSON)
    for k in range(4):
        dn.add_lattice(k, side, side)
        dn.add_spike_train_lattice(4 + k, side, side)
    if shard is None:
        dn.finalize(csr=True)
    else:
        dn.finalize(shard[0], shard[1], csr=True, by_lattice=by_lattice)
    for k in range(4):
        dn.set_attr(k, "gap_conductance", np.full(m, 10.0, f32))
        dn.set_attr(k, "current_voltage", v0[k * m:(k + 1) * m])
        dn.set_attr(4 + k, "chance_of_firing", np.full(m, 0.01, f32))
        dn.set_attr(4 + k, "seed", np.arange(k * m + 1, (k + 1) * m + 1, dtype=np.uint32))
    dn.set_graph_csr(*synthetic.c5_csr(side, posts=dn.owned))
    return dn


def test_c5_full_size_sparse_network_against_oracle_and_numpy(snn):
    """BASELINE configs[4] at full size: 1 048 576 neurons, 1 048 576 Poisson cells, 14.6 M synapses, 30 steps.
    Voltages, adaptation variables, spikes, firing times and the cells' generator state bit-identical to the C oracle's
    sparse routine (snn_o_run_csr) AND to a numpy restatement of the same sparse step (gap junctions from neurons and from Poisson cells with the delta-dirac
    refractoriness, the canonical 256-chunk summation order, Izhikevich update, xorshift32 cells) -- on ONE handle and
    on EIGHT shard handles (configs[4]'s multi-GPU shape on one device) that trade halo segments: per handle the
    two lattice rows either side of its slab and the half lattice the ring edge k -> k + 1 reads, not whole slots."""
    import torch
    from snn_amd import parallel, synthetic
    import numpy_ref as nr
    f32 = np.float32
    side, steps = 512, 30
    m = side * side
    nn = nc = 4 * m
    ptr, pre, w = synthetic.c5_csr(side)
    v0 = np.concatenate([synthetic.uniform(6, m, -65.0, 30.0, offset=k * m) for k in range(4)])
    dn = c5_handle(snn, side, v0)
    dn.run(steps)

    # ---- numpy restatement ----
    deg = np.diff(ptr).astype(np.int64)
    K = int(deg.max())
    idx = np.full((nn, K), -1, np.int64)
    rows = np.repeat(np.arange(nn), deg)
    idx[rows, np.arange(pre.size) - np.repeat(ptr[:-1].astype(np.int64), deg)] = pre
    n_in = deg.astype(f32)
    st = {"current_voltage": v0.copy(), "w_value": np.full(nn, 30.0, f32), "a": np.full(nn, 0.02, f32),
          "b": np.full(nn, 0.2, f32), "c": np.full(nn, -55.0, f32), "d": np.full(nn, 8.0, f32),
          "v_th": np.full(nn, 30.0, f32), "tau_m": np.full(nn, 1.0, f32), "c_m": np.full(nn, 100.0, f32),
          "dt": np.full(nn, 0.1, f32)}
    g = f32(10.0)
    lft = np.full(nn, -1, np.int32)
    seed = np.arange(1, nc + 1, dtype=np.uint32)
    st_lft = np.full(nc, -1, np.int32)
    expf = np.vectorize(ob.expf, otypes=[np.float32])
    total_spikes = 0
    for t in range(steps):
        # presynaptic value of the cells at clock t (spike_train_gap_junction, neuron/mod.rs:119-137)
        fired = st_lft >= 0
        eff = np.zeros(nc, f32)
        td = (t - st_lft[fired]).astype(f32)
        scale = f32(f32(-1.0) / f32(f32(10000.0) / f32(0.1)))
        eff[fired] = (f32(30.0) * expf((scale * (td * td).astype(f32)).astype(f32))).astype(f32) + f32(0.0)
        v = st["current_voltage"]
        total = np.zeros(nn, f32)
        part = np.zeros(nn, f32)
        cur = np.full(nn, -1, np.int64)
        for j in range(K):
            p = idx[:, j]
            valid = p >= 0
            pc = np.where(valid, p, 0)
            is_cell = pc >= nn
            cell = np.where(is_cell, pc - nn, 0)
            term_n = (g * (v[np.where(is_cell, 0, pc)] - v).astype(f32)).astype(f32)
            term_c = np.where(fired[cell], (g * eff[cell]).astype(f32), f32(0.0)).astype(f32)   # never fired: v_resting
            term = np.where(is_cell, term_c, term_n).astype(f32)
            chunk = pc // 256
            flush = valid & (chunk != cur) & (cur >= 0)
            total = np.where(flush, (total + part).astype(f32), total)
            part = np.where(valid & (chunk != cur), f32(0.0), part)
            cur = np.where(valid, chunk, cur)
            part = np.where(valid, (part + (term * f32(1.0)).astype(f32)).astype(f32), part)
        total = np.where(cur >= 0, (total + part).astype(f32), total)
        i_in = (total / n_in).astype(f32)
        spike = nr.izhikevich_step(st, i_in)
        lft[spike] = t
        total_spikes += int(spike.sum())
        # Poisson cells (GPU generator of the reference, spike_train/mod.rs:380-388, 419-426)
        with np.errstate(over="ignore"):
            seed ^= seed << np.uint32(13)
            seed ^= seed >> np.uint32(17)
            seed ^= seed << np.uint32(5)
        cs = (seed.astype(f32) / f32(4294967296.0)).astype(f32) < f32(0.01)
        st_lft[cs] = t
    assert total_spikes > 100 and (st_lft >= 0).sum() > 100_000

    # ---- the C oracle over the same CSR rows (snn_o_run_csr: the pinned restatement, canonical chunk flush) ----
    net = ob.Net(nn, model=ob.IZHIKEVICH, n_cells=nc, st_kind=ob.ST_POISSON, n_lattices=4, n_st_lattices=4, dense=False)
    net["lattice"] = np.repeat(np.arange(4, dtype=np.uint32), m)
    net["st_lattice"] = np.repeat(np.arange(4, dtype=np.uint32), m)
    net["current_voltage"] = v0
    net["gap_conductance"] = 10.0
    net["st_chance_of_firing"] = 0.01
    net["st_seed"] = np.arange(1, nc + 1, dtype=np.uint32)
    net.n_threads = ob.usable_cpus()
    net.run_csr(np.ascontiguousarray(ptr, np.uint64), np.ascontiguousarray(pre, np.uint32),
                np.ascontiguousarray(w, np.float32), steps)
    oracle = {"current_voltage": net["current_voltage"], "w_value": net["w_value"], "lft": net["last_firing_time"],
              "seed": net["st_seed"], "st_lft": net["st_last_firing_time"]}
    # the two CPU restatements agree with each other ...
    assert np.array_equal(parity.bits(oracle["current_voltage"]), parity.bits(st["current_voltage"]))
    assert np.array_equal(parity.bits(oracle["w_value"]), parity.bits(st["w_value"]))
    assert np.array_equal(oracle["lft"], lft) and np.array_equal(oracle["seed"], seed) and np.array_equal(oracle["st_lft"], st_lft)
    # ... and the device with the oracle
    for k in range(4):
        sl = slice(k * m, (k + 1) * m)
        assert np.array_equal(parity.bits(dn.get_attr(k, "current_voltage")), parity.bits(oracle["current_voltage"][sl])), k
        assert np.array_equal(parity.bits(dn.get_attr(k, "w_value")), parity.bits(oracle["w_value"][sl])), k
        assert np.array_equal(dn.get_attr(k, "last_firing_time", dtype=np.int32), oracle["lft"][sl]), k
        assert np.array_equal(dn.get_attr(4 + k, "seed", dtype=np.uint32), oracle["seed"][sl]), k
        assert np.array_equal(dn.get_attr(4 + k, "last_firing_time", dtype=np.int32), oracle["st_lft"][sl]), k
    del net

    for k in range(4):
        sl = slice(k * m, (k + 1) * m)
        assert np.array_equal(parity.bits(dn.get_attr(k, "current_voltage")), parity.bits(st["current_voltage"][sl])), k
        assert np.array_equal(parity.bits(dn.get_attr(k, "w_value")), parity.bits(st["w_value"][sl])), k
        assert np.array_equal(dn.get_attr(k, "last_firing_time", dtype=np.int32), lft[sl]), k
        assert np.array_equal(dn.get_attr(4 + k, "seed", dtype=np.uint32), seed[sl]), k
        assert np.array_equal(dn.get_attr(4 + k, "last_firing_time", dtype=np.int32), st_lft[sl]), k
    dn.close()

    # ---- eight shard handles on this GPU, halo exchange ----
    g = 8
    handles = [c5_handle(snn, side, v0, shard=(r, g)) for r in range(g)]
    ex = parallel.LocalExchange(handles, torch.device("cuda", 0), halo=True)
    assert all(p["mode"] == "halo" and p["plane_id"] == [0] for p in ex.plans)
    # per handle and step: the half lattice behind the ring edge (131 072 neurons) + 2 x 2 lattice rows of 512, at
    # 4 B + 1 bit each -- 0.54 MB instead of the 4.3 MB of whole slots (and of the 21 MB of the five-plane all-gather
    # this replaces)
    per_step = ex.bytes_per_step()
    assert max(per_step) <= 4 * ((m // 2 + 4 * side) + (m // 2 + 4 * side) // 32 + 8), per_step
    for _ in range(steps):
        ex.step()
    for h in handles:
        b, e = h.post_begin, h.post_end
        for k in range(4):
            lo, hi = max(b, k * m), min(e, (k + 1) * m)
            if lo >= hi:
                continue
            own = slice(lo - k * m, hi - k * m)
            assert np.array_equal(parity.bits(h.get_attr(k, "current_voltage")[own]), parity.bits(st["current_voltage"][lo:hi])), k
            assert np.array_equal(parity.bits(h.get_attr(k, "w_value")[own]), parity.bits(st["w_value"][lo:hi])), k
            assert np.array_equal(h.get_attr(k, "last_firing_time", dtype=np.int32)[own], lft[lo:hi]), k
            # the cells this handle's rows read (one per own neuron) are current
            assert np.array_equal(h.get_attr(4 + k, "seed", dtype=np.uint32)[own], seed[lo:hi]), k
            assert np.array_equal(h.get_attr(4 + k, "last_firing_time", dtype=np.int32)[own], st_lft[lo:hi]), k
        # what the handle reads of the others: voltage and firing times of its halo
        r = ex.plans[handles.index(h)]["shard_index"]
        for p in range(g):
            need = h.halo_needs(p) if p != r else np.zeros(0, np.uint32)
            if need.size:
                v_all = np.concatenate([h.get_attr(k, "current_voltage") for k in range(4)])
                l_all = np.concatenate([h.get_attr(k, "last_firing_time", dtype=np.int32) for k in range(4)])
                assert np.array_equal(parity.bits(v_all[need]), parity.bits(st["current_voltage"][need])), (r, p)
                assert np.array_equal(l_all[need], lft[need]), (r, p)
                break                           # one peer per handle keeps the test short
        h.close()

    # ---- eight shard handles that own the same slab of EVERY lattice: the ring edge k -> k+1 stays inside a handle ----
    handles = [c5_handle(snn, side, v0, shard=(r, g), by_lattice=True) for r in range(g)]
    ex = parallel.LocalExchange(handles, torch.device("cuda", 0), halo=True)
    # per handle and step: two lattice rows of 512 either side of each of its four slabs (fewer at the lattice edges),
    # 4 B + 1 bit each: <= 33.8 KB instead of 545 KB
    per_step = ex.bytes_per_step()
    assert max(per_step) <= 4 * (4 * 4 * side + 4 * 4 * side // 32 + 8), per_step
    for _ in range(steps):
        ex.step()
    for h in handles:
        o = h.owned
        assert len(h.ranges) == 4 and o.size == nn // g
        v_all = np.concatenate([h.get_attr(k, "current_voltage") for k in range(4)])
        w_all = np.concatenate([h.get_attr(k, "w_value") for k in range(4)])
        l_all = np.concatenate([h.get_attr(k, "last_firing_time", dtype=np.int32) for k in range(4)])
        assert np.array_equal(parity.bits(v_all[o]), parity.bits(st["current_voltage"][o]))
        assert np.array_equal(parity.bits(w_all[o]), parity.bits(st["w_value"][o]))
        assert np.array_equal(l_all[o], lft[o])
        r = ex.plans[handles.index(h)]["shard_index"]
        need = np.concatenate([h.halo_needs(p) for p in range(g) if p != r])
        assert np.array_equal(parity.bits(v_all[need]), parity.bits(st["current_voltage"][need]))
        assert np.array_equal(l_all[need], lft[need])
        cells = h.cells_read()
        assert np.array_equal(cells, o)                      # one Poisson cell per own neuron, nothing else
        seed_all = np.concatenate([h.get_attr(4 + k, "seed", dtype=np.uint32) for k in range(4)])
        assert np.array_equal(seed_all[cells], seed[cells])
        h.close()
