"""Shared parity harness: build the SAME network for the CPU oracle and the HIP stepper,
run both, compare every state array bit for bit.  Test infrastructure only."""
import numpy as np

import oracle_binding as ob

TYPE_NAMES = ("AMPA", "NMDA", "GABA")

# oracle array name -> C-ABI attribute name (reference buffer names), per model
COMMON = {"current_voltage": "current_voltage", "gap_conductance": "gap_conductance", "dt": "dt",
          "c_m": "c_m", "v_th": "v_th", "is_spiking": "is_spiking", "last_firing_time": "last_firing_time"}
MODEL_ATTRS = {
    ob.IZHIKEVICH: {k: k for k in ("w_value", "a", "b", "c", "d", "tau_m")},
    ob.LIF: {k: k for k in ("tau_m", "v_reset", "refractory_count", "tref", "leak_constant",
                            "integration_constant", "e_l", "g_l")},
    ob.QIF: {"qif_alpha": "alpha", "qif_v_c": "v_c", "v_reset": "v_reset", "refractory_count": "refractory_count",
             "tref": "tref", "integration_constant": "integration_constant", "tau_m": "tau_m"},
    ob.SIMPLE_LIF: {"slif_g": "g", "slif_e": "e", "v_reset": "v_reset"},
    ob.ADAPTIVE_LIF: {**{k: k for k in ("tau_m", "v_reset", "refractory_count", "tref", "leak_constant",
                                        "integration_constant", "e_l", "g_l", "w_value")},
                      "adp_alpha": "alpha", "adp_beta": "beta"},
    ob.ADAPTIVE_EXP_LIF: {**{k: k for k in ("tau_m", "v_reset", "refractory_count", "tref", "leak_constant",
                                            "integration_constant", "e_l", "g_l", "w_value", "slope_factor")},
                          "adp_alpha": "alpha", "adp_beta": "beta"},
    ob.LEAKY_IZHIKEVICH: {k: k for k in ("w_value", "a", "b", "c", "d", "tau_m", "e_l")},
    ob.BCM_IZHIKEVICH: {**{k: k for k in ("w_value", "a", "b", "c", "d", "tau_m")},
                        "bcm_average_activity": "average_activity", "bcm_current_activity": "current_activity",
                        "bcm_clock": "firing_rate_clock", "bcm_window": "firing_rate_window", "bcm_period": "period",
                        "bcm_num_spikes": "num_spikes"},
    ob.HH: {"m_state": "na_channel$m$state", "h_state": "na_channel$h$state", "n_state": "k_channel$n$state",
            "m_alpha": "na_channel$m$alpha", "m_beta": "na_channel$m$beta",
            "h_alpha": "na_channel$h$alpha", "h_beta": "na_channel$h$beta",
            "n_alpha": "k_channel$n$alpha", "n_beta": "k_channel$n$beta",
            "g_na": "na_channel$g_na", "e_na": "na_channel$e_na", "g_k": "k_channel$g_k", "e_k": "k_channel$e_k",
            "g_k_leak": "k_leak_channel$g_k_leak", "e_k_leak": "k_leak_channel$e_k_leak",
            "na_current": "na_channel$current", "k_current": "k_channel$current",
            "k_leak_current": "k_leak_channel$current", "was_increasing": "was_increasing"},
}
NT_ATTRS = {"nt_t": "neurotransmitters$t", "nt_t_max": "neurotransmitters$t_max",
            "nt_clearance": "neurotransmitters$clearance_constant", "nt_v_p": "neurotransmitters$v_p",
            "nt_k_p": "neurotransmitters$k_p", "nt_flags": "neurotransmitters$flags"}
RC_PER_TYPE = {"rc_g": "receptors${T}_g", "rc_e": "receptors${T}_e", "rc_current": "receptors${T}_current",
               "rc_r": "receptors${T}$r$kinetics$r", "rc_alpha": "receptors${T}$r$kinetics$alpha",
               "rc_beta": "receptors${T}$r$kinetics$beta"}
CELL_ATTRS = {"st_current_voltage": "current_voltage", "st_v_th": "v_th", "st_v_resting": "v_resting",
              "st_dt": "dt", "st_k": "neural_refractoriness$k", "st_refractoriness": "neural_refractoriness$kind",
              "st_is_spiking": "is_spiking",
              "st_last_firing_time": "last_firing_time"}
CELL_KIND_ATTRS = {ob.ST_POISSON: {"st_chance_of_firing": "chance_of_firing", "st_seed": "seed"},
                   ob.ST_RATE: {"st_rate": "rate", "st_step": "step"},
                   ob.ST_PRESET: {"st_step": "internal_clock", "st_counter": "counter"},
                   ob.ST_BCM_POISSON: {"st_chance_of_firing": "chance_of_firing", "st_seed": "seed",
                                       "st_bcm_average_activity": "average_activity",
                                       "st_bcm_current_activity": "current_activity", "st_bcm_clock": "firing_rate_clock",
                                       "st_bcm_window": "firing_rate_window", "st_bcm_period": "period",
                                       "st_bcm_num_spikes": "num_spikes"}}


class Layout:
    """Lattice ids and shapes of a network: neuron lattices then spike-train lattices (ascending id)."""

    def __init__(self, lattices, st_lattices=()):
        self.lattices = sorted(lattices)          # (id, rows, cols)
        self.st_lattices = sorted(st_lattices)
        self.n_neurons = sum(r * c for _, r, c in self.lattices)
        self.n_cells = sum(r * c for _, r, c in self.st_lattices)

    def ranges(self):
        off = 0
        out = {}
        for i, r, c in self.lattices:
            out[i] = (off, r * c, False)
            off += r * c
        off = 0
        for i, r, c in self.st_lattices:
            out[i] = (off, r * c, True)
            off += r * c
        return out


def make_oracle(layout, **kw):
    net = ob.Net(layout.n_neurons, n_cells=layout.n_cells, n_lattices=max(1, len(layout.lattices)),
                 n_st_lattices=len(layout.st_lattices), **kw)
    rng = layout.ranges()
    net["lattice_count"][...] = 0
    for slot, (i, r, c) in enumerate(layout.lattices):
        first, count, _ = rng[i]
        net["lattice"][first:first + count] = slot
        net["lattice_first"][slot], net["lattice_count"][slot] = first, count
    for slot, (i, r, c) in enumerate(layout.st_lattices):
        first, count, _ = rng[i]
        net["st_lattice"][first:first + count] = slot
    net.layout = layout
    return net


def csr_from_dense(net, q0, q1):
    """CSR (by postsynaptic neuron q0..q1) of the oracle's dense masked matrix: ascending presynaptic index."""
    conn = net["connections"][:, q0:q1]
    pre, post = np.nonzero(conn.T)[::-1]                 # iterate posts (rows of conn.T) in order, pres ascending
    counts = conn.sum(axis=0, dtype=np.uint64)
    row_ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    weights = net["weights"][:, q0:q1].T[conn.T != 0]
    return row_ptr, pre.astype(np.uint32), weights.astype(np.float32)


def csr_for_posts(net, posts):
    """CSR rows of the postsynaptic neurons `posts` (ascending global indices) of the oracle's dense masked matrix"""
    posts = np.asarray(posts, dtype=np.int64)
    conn = net["connections"][:, posts]
    pre, _ = np.nonzero(conn.T)[::-1]
    counts = conn.sum(axis=0, dtype=np.uint64)
    row_ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    weights = net["weights"][:, posts].T[conn.T != 0]
    return row_ptr, pre.astype(np.uint32), weights.astype(np.float32)


def dense_from_csr(net, q0, q1, csr_weights):
    """Scatter CSR-ordered weights back into a dense [n_tot, q1-q0] block (absent edges 0)."""
    conn = net["connections"][:, q0:q1]
    out = np.zeros(conn.T.shape, np.float32)
    out[conn.T != 0] = csr_weights
    return out.T


def device_from_oracle(snn, net, shard=None, device=0, csr=False, by_lattice=False):
    """Create a DeviceNetwork holding exactly the oracle net's state (dense or CSR graph form).  shard = (index, count);
    by_lattice: the shard owns that slab of every lattice (sparse handles)."""
    lay = net.layout
    dn = snn.DeviceNetwork(model=net.model, nt_kinetics=net.nt_kind, receptor_kinetics=net.rc_kind,
                           spike_train=net.st_kind, device=device, lib_path=getattr(net, "custom_lib", None))
    for i, r, c in lay.lattices:
        dn.add_lattice(i, r, c)
    for i, r, c in lay.st_lattices:
        dn.add_spike_train_lattice(i, r, c)
    if shard is None:
        dn.finalize(csr=csr)
    else:
        dn.finalize(*shard, csr=csr, by_lattice=by_lattice)
    push_state(dn, net)
    if net.st_kind == ob.ST_PRESET:
        rng = lay.ranges()
        ptr, times = net["st_firing_ptr"], net["st_firing_times"]
        for i, r, c in lay.st_lattices:
            first, count, _ = rng[i]
            lo, hi = int(ptr[first]), int(ptr[first + count])
            dn.set_firing_times(i, ptr[first:first + count + 1] - ptr[first], times[lo:hi])
    nn = net.n_neurons
    if csr:
        dn.set_graph_csr(*csr_for_posts(net, dn.owned))
    elif net.n_tot and nn:
        dn.set_graph_rows(0, net["weights"], net["connections"].astype(np.uint32))
    dn.set_synapses(net.electrical, net.chemical)
    for slot, (i, r, c) in enumerate(lay.lattices):
        if net["plasticity_kind"][slot]:
            dn.set_bcm(i, float(net["bcm_decay"][slot]), float(net["bcm_average_scalar"][slot]),
                       float(net["bcm_dt"][slot]), bool(net["do_plasticity"][slot]))
            continue
        dn.set_plasticity(i, float(net["stdp_a_plus"][slot]), float(net["stdp_a_minus"][slot]),
                          float(net["stdp_tau_plus"][slot]), float(net["stdp_tau_minus"][slot]),
                          float(net["stdp_dt"][slot]), bool(net["do_plasticity"][slot]))
    return dn


def push_connection_kinds(dn, net):
    """the connections of a reward-modulated network (conn_kind / pending / edge_counter of the oracle net) onto a device handle;
    call after the reward modulators and traces are set"""
    kinds = net["conn_kind"]
    if not kinds.any():
        return
    lay = net.layout
    sources = [i for i, _, _ in lay.lattices] + [i for i, _, _ in lay.st_lattices]
    for s, pre in enumerate(sources):
        for l, (post, _, _) in enumerate(lay.lattices):
            if kinds[s, l]:
                dn.set_connection_kind(pre, post, int(kinds[s, l]))
    if getattr(dn, "csr", False):
        dn.set_pending_csr(csr_values(net, net["pending"], dn.owned))
        dn.set_counters_csr(csr_values(net, net["edge_counter"], dn.owned))
        return
    dn.set_pending_rows(0, net["pending"])
    dn.set_counter_rows(0, net["edge_counter"])


def csr_values(net, dense, posts):
    """values of a dense [n_tot][n_neurons] array in the CSR-by-post edge order of the columns `posts`"""
    ptr, pre, _ = csr_for_posts(net, posts)
    return (np.concatenate([dense[pre[ptr[k]:ptr[k + 1]], q] for k, q in enumerate(posts)]) if len(posts)
            else np.zeros(0, dense.dtype))


def _neuron_names(net):
    names = dict(COMMON)
    if net.model == ob.CUSTOM:
        # a DSL variable may carry a common name (v_th): the generated model's attribute replaces it
        for name, _ in net.custom_model.variables:
            names = {o: a for o, a in names.items() if a != name}
        return names
    names.update(MODEL_ATTRS[net.model])
    return names


def _nt_attrs(net):
    """built-in neurotransmitter attributes that a generated kinetics variable of the same name does not replace"""
    taken = {"neurotransmitters$" + name for name in _kinetics_names(net, "nt_model")}
    return {o: a for o, a in NT_ATTRS.items() if a not in taken}


def _rc_per_type(net):
    """built-in per-type receptor attributes that a generated receptor-kinetics variable does not replace"""
    taken = {"$r$kinetics$" + name for name in _kinetics_names(net, "rc_model")}
    return {o: pat for o, pat in RC_PER_TYPE.items() if not any(pat.endswith(t) for t in taken)}


def _rc_types(net, pattern):
    """(type index, built-in type name) pairs whose attribute `pattern` is NOT a variable of the generated receptor set
    (a set may have a type that carries a built-in type's name: the attribute then means the set's variable)"""
    own = {"receptors$" + name for name in _kinetics_names(net, "rx_model")}
    return [(k, t) for k, t in enumerate(TYPE_NAMES) if pattern.replace("{T}", t) not in own]


def _kinetics_names(net, which):
    model = getattr(net, which, None)
    return [name for name, _ in model.variables] if model is not None else []


def push_state(dn, net):
    rng = net.layout.ranges()
    for i, (first, count, is_st) in rng.items():
        if count == 0:
            continue
        sl = slice(first, first + count)
        if not is_st:
            for o, a in _neuron_names(net).items():
                dn.set_attr(i, a, net[o][sl])
            if net.model == ob.CUSTOM:
                for k, (name, _) in enumerate(net.custom_model.variables):
                    dn.set_attr(i, name, np.ascontiguousarray(net["custom_vars"][k, sl]))
            for o, a in _nt_attrs(net).items():
                dn.set_attr(i, a, net[o][sl])
            for k, name in enumerate(_kinetics_names(net, "nt_model")):
                dn.set_attr(i, "neurotransmitters$" + name, np.ascontiguousarray(net["nt_custom_vars"][k, sl]))
            for k, name in enumerate(_kinetics_names(net, "rx_model")):
                dn.set_attr(i, "receptors$" + name, np.ascontiguousarray(net["rx_vars"][k, sl]))
            for k, name in enumerate(_kinetics_names(net, "rc_model")):
                for ty, t in _rc_types(net, "receptors${T}$r$kinetics$" + name):
                    dn.set_attr(i, f"receptors${t}$r$kinetics${name}", np.ascontiguousarray(net["rc_custom_vars"][k, sl, ty]))
            dn.set_attr(i, "receptors$flags", net["rc_flags"][sl])
            for o, pat in _rc_per_type(net).items():
                for k, t in _rc_types(net, pat):
                    dn.set_attr(i, pat.replace("{T}", t), np.ascontiguousarray(net[o][sl, k]))
            dn.set_attr(i, "receptors$NMDA_mg", np.ascontiguousarray(net["rc_mg"][sl, 1]))
        else:
            for o, a in CELL_ATTRS.items():
                dn.set_attr(i, a, net[o][sl])
            for o, a in CELL_KIND_ATTRS.get(net.st_kind, {}).items():
                dn.set_attr(i, a, net[o][sl])
            if net.st_kind == ob.ST_CUSTOM:
                for k, (name, _) in enumerate(net.st_custom_model.variables):
                    dn.set_attr(i, name, np.ascontiguousarray(net["st_custom_vars"][k, sl]))
            if getattr(net, "refr_model", None) is not None:
                for k, (name, _) in enumerate(net.refr_model.variables):
                    dn.set_attr(i, "neural_refractoriness$" + name, np.ascontiguousarray(net["refr_vars"][k, sl]))
            for o, a in _nt_attrs(net).items():
                dn.set_attr(i, a, net["st_" + o][sl])
            for k, name in enumerate(_kinetics_names(net, "nt_model")):
                dn.set_attr(i, "neurotransmitters$" + name, np.ascontiguousarray(net["st_nt_custom_vars"][k, sl]))


def pull_state(dn, net):
    """Download the device state into a dict keyed like the oracle's arrays."""
    out = {}
    rng = net.layout.ranges()

    def put(name, sl, val):
        if name not in out:
            out[name] = np.zeros_like(net[name])
        out[name][sl] = val.reshape(out[name][sl].shape)

    for i, (first, count, is_st) in rng.items():
        if count == 0:
            continue
        sl = slice(first, first + count)
        if not is_st:
            for o, a in _neuron_names(net).items():
                put(o, sl, dn.get_attr(i, a, dtype=net[o].dtype))
            if net.model == ob.CUSTOM:
                if "custom_vars" not in out:
                    out["custom_vars"] = np.zeros_like(net["custom_vars"])
                for k, (name, _) in enumerate(net.custom_model.variables):
                    out["custom_vars"][k, sl] = dn.get_attr(i, name)
            for o, a in _nt_attrs(net).items():
                put(o, sl, dn.get_attr(i, a, dtype=net[o].dtype, per_type=True))
            for k, name in enumerate(_kinetics_names(net, "nt_model")):
                if "nt_custom_vars" not in out:
                    out["nt_custom_vars"] = np.zeros_like(net["nt_custom_vars"])
                out["nt_custom_vars"][k, sl] = dn.get_attr(i, "neurotransmitters$" + name, per_type=True).reshape(-1, 3)
            for k, name in enumerate(_kinetics_names(net, "rx_model")):
                if "rx_vars" not in out:
                    out["rx_vars"] = np.zeros_like(net["rx_vars"])
                out["rx_vars"][k, sl] = dn.get_attr(i, "receptors$" + name)
            for k, name in enumerate(_kinetics_names(net, "rc_model")):
                if "rc_custom_vars" not in out:
                    out["rc_custom_vars"] = np.zeros_like(net["rc_custom_vars"])
                out["rc_custom_vars"][k, sl] = net["rc_custom_vars"][k, sl]
                for ty, t in _rc_types(net, "receptors${T}$r$kinetics$" + name):
                    out["rc_custom_vars"][k, sl, ty] = dn.get_attr(i, f"receptors${t}$r$kinetics${name}")
            put("rc_flags", sl, dn.get_attr(i, "receptors$flags", dtype=np.uint32, per_type=True))
            for o, pat in _rc_per_type(net).items():
                if o not in out:
                    out[o] = net[o].copy()
                for k, t in _rc_types(net, pat):
                    out[o][sl, k] = dn.get_attr(i, pat.replace("{T}", t))
        else:
            for o, a in CELL_ATTRS.items():
                put(o, sl, dn.get_attr(i, a, dtype=net[o].dtype))
            for o, a in CELL_KIND_ATTRS.get(net.st_kind, {}).items():
                put(o, sl, dn.get_attr(i, a, dtype=net[o].dtype))
            if net.st_kind == ob.ST_CUSTOM:
                if "st_custom_vars" not in out:
                    out["st_custom_vars"] = np.zeros_like(net["st_custom_vars"])
                for k, (name, _) in enumerate(net.st_custom_model.variables):
                    out["st_custom_vars"][k, sl] = dn.get_attr(i, name)
            for o, a in _nt_attrs(net).items():
                put("st_" + o, sl, dn.get_attr(i, a, dtype=net["st_" + o].dtype, per_type=True))
            for k, name in enumerate(_kinetics_names(net, "nt_model")):
                if "st_nt_custom_vars" not in out:
                    out["st_nt_custom_vars"] = np.zeros_like(net["st_nt_custom_vars"])
                out["st_nt_custom_vars"][k, sl] = dn.get_attr(i, "neurotransmitters$" + name, per_type=True).reshape(-1, 3)
    return out


def bits(a):
    """Bit pattern of a float32 array for exact comparison.  Every NaN maps to one pattern: IEEE 754 leaves the
    sign / payload of a propagated NaN open when both operands are NaN -- x86 SSE returns the first operand,
    gfx950 does not (measured: (+qNaN) + (-qNaN) gives 7fc00000 on the host, ffc00000 on the device) -- and Rust
    makes no promise about NaN bits either.  Finite values, infinities and signed zeros stay bit-exact."""
    a = np.ascontiguousarray(a)
    if a.dtype != np.float32:
        return a
    u = a.view(np.uint32).copy()
    u[np.isnan(a)] = 0x7FC00000
    return u


def assert_state_equal(net, dev_state, skip=()):
    """Bit-exact comparison of every downloaded array (NaNs compare equal to NaNs, see bits())."""
    bad = []
    for name, dv in dev_state.items():
        if name in skip:
            continue
        ov = net[name]
        if not np.array_equal(bits(ov), bits(dv)):
            idx = np.argwhere(bits(ov) != bits(dv))
            first = tuple(idx[0])
            bad.append(f"{name}: {len(idx)} mismatches, first at {first}: oracle={ov[first]!r} hip={dv[first]!r}")
    assert not bad, "state differs from the oracle:\n  " + "\n  ".join(bad)


def assert_graph_equal(net, dn):
    if net.n_tot == 0 or net.n_neurons == 0:
        return
    oc = net["connections"].astype(np.uint32)
    if getattr(dn, "csr", False):
        _, _, want = csr_for_posts(net, dn.owned)
        assert np.array_equal(bits(want), bits(dn.get_graph_csr())), "CSR weights differ"
        return
    w, c = dn.get_graph_rows(0, net.n_tot)
    assert np.array_equal(c, oc), "connection masks differ"
    ow = np.where(oc != 0, net["weights"], np.float32(0))
    if not np.array_equal(bits(ow), bits(w)):
        idx = np.argwhere(bits(ow) != bits(w))
        f = tuple(idx[0])
        raise AssertionError(f"weights differ at {len(idx)} places, first {f}: oracle={ow[f]!r} hip={w[f]!r}")


def assert_shard_view_equal(h, st, net):
    """The part of the OTHER shards' state a shard handle holds after snn_step_end -- the planes of its exchange plan
    (voltage with gap junctions, t of the transmitter types in use with chemical synapses, the spike bit and with it
    last_firing_time), for every neuron (all-gather) or for the neurons its rows read (halo) -- and all of its own."""
    plan = h.exchange_plan()
    nn = net.n_neurons
    known = np.zeros(nn, bool)
    own = np.zeros(nn, bool)
    own[h.owned] = True
    if plan["mode"] == "halo":
        for p in range(plan["n_shards"]):
            if p != plan["shard_index"]:
                known[h.halo_needs(p)] = True
    else:
        known[:] = True
    known |= own
    for name in ("is_spiking", "last_firing_time"):
        assert np.array_equal(bits(st[name][known]), bits(net[name][known])), name
    v = known if 0 in plan["plane_id"] else own
    assert np.array_equal(bits(st["current_voltage"][v]), bits(net["current_voltage"][v])), "current_voltage"
    for k in range(3):
        m = known if (2 + k) in plan["plane_id"] else own
        assert np.array_equal(bits(st["nt_t"][m, k]), bits(net["nt_t"][m, k])), f"nt_t type {k}"
