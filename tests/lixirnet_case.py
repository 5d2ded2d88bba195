"""The neuron of the reference's GPU Python module (interface_gpu/lixirnet/src/lib.rs:22-79; snn_amd.examples_dsl.LIXIRNET)
as test infrastructure: `oracle_net` lays out an oracle network of it (the description compiled to the oracle's stack
programs), `LixirnetTwin` is a second, hand-written numpy float32 restatement of ONE network step of exactly this
description -- formulas typed in from the DSL text, libm's expf / powf through tests/numpy_net.py -- used to check the
generator's reading of the description (operator order, the two receptor states of glutamate, `r^`, the dopamine
modifiers acting one step later)."""
import numpy as np

import modelgen_ref
import numpy_net
import oracle_binding as ob
import parity

f32 = np.float32
GLU, GABA, DOPA = 0, 1, 2


def description():
    from snn_amd import modelgen
    from snn_amd.examples_dsl import LIXIRNET
    return modelgen.parse_description(LIXIRNET)


def oracle_net(layout, electrical=True, chemical=False, st_kind=ob.ST_NONE):
    desc = description()
    net = parity.make_oracle(layout, model=ob.CUSTOM, nt_kind=ob.NT_CUSTOM, rc_kind=ob.RC_CUSTOM, st_kind=st_kind,
                             electrical=electrical, chemical=chemical)
    modelgen_ref.attach(net, desc.neuron)
    modelgen_ref.attach_nt_kinetics(net, desc.nt_kinetics)
    modelgen_ref.attach_receptor_kinetics(net, desc.receptor_kinetics)
    modelgen_ref.attach_receptors(net, desc.receptors)
    net.description = desc
    return net


def var(net, table, name):
    """the array of a generated variable: table = custom_vars / rx_vars / nt_custom_vars / st_nt_custom_vars"""
    d = net.description
    names = {"custom_vars": d.neuron.variables, "rx_vars": d.receptors.variables, "nt_custom_vars": d.nt_kinetics.variables,
             "st_nt_custom_vars": d.nt_kinetics.variables}[table]
    return net[table][[n for n, _ in names].index(name)]


class LixirnetTwin(numpy_net.NumpyNet):
    """inputs / spike trains / loop of NumpyNet; the neuron, its transmitter kinetics and its receptor set restated here"""

    def __init__(self, src):
        super().__init__(src)
        d = src.description
        self.nv = {n: k for k, (n, _) in enumerate(d.neuron.variables)}
        self.rv = {n: k for k, (n, _) in enumerate(d.receptors.variables)}
        self.kv = {n: k for k, (n, _) in enumerate(d.nt_kinetics.variables)}

    # BoundedNeurotransmitterKinetics, every type a cell carries
    def nt_apply(self, prefix, voltage, spiking, dt):
        a = self.a
        x = a[prefix + "nt_custom_vars"]
        for k in range(3):
            on = a[prefix + "nt_flags"][:, k] != 0
            if not on.any():
                continue
            t_max, cc = x[self.kv["t_max"]][on, k], x[self.kv["clearance_constant"]][on, k]
            conc = np.where(spiking[on], t_max, f32(0.0)).astype(f32)
            t = a[prefix + "nt_t"][on, k]
            t = ((t + ((dt[on] * (-cc)).astype(f32) * t).astype(f32)).astype(f32) + conc).astype(f32)
            t = numpy_net.rust_min(numpy_net.rust_max(t, f32(0.0)), t_max)
            x[self.kv["conc"]][on, k] = conc
            a[prefix + "nt_t"][on, k] = t

    def update_neurons(self, i_in, t_in, t_cnt):
        a, R = self.a, self.a["rx_vars"]
        rv = lambda name: R[self.rv[name]]
        v = a["current_voltage"].copy()
        dt, c_m = a["dt"], a["c_m"]
        C = a["custom_vars"]
        u, pa, pb, pc, pd, v_th, tau_m = (C[self.nv[k]] for k in ("u", "a", "b", "c", "d", "v_th", "tau_m"))
        prev_spiking = a["is_spiking"] != 0
        have = a["rc_flags"] != 0
        with np.errstate(all="ignore"):
            if self.chemical:
                # BoundedReceptorKinetics on every state of a type present in the input and in the set
                for k, states in ((GLU, ("Glutamate$ampa_r", "Glutamate$nmda_r")), (GABA, ("GABA$r",)),
                                  (DOPA, ("Dopamine$r_d1", "Dopamine$r_d2"))):
                    upd = have[:, k] & (t_cnt[:, k] != 0)
                    for s in states:
                        r, r_max = rv(s + "$kinetics$r"), rv(s + "$kinetics$r_max")
                        r[upd] = numpy_net.rust_min(numpy_net.rust_max(t_in[upd, k], f32(0.0)), r_max[upd])
                # the receptors present iterate in declaration order at the OLD voltage
                inh, nmod = rv("inh_modifier"), rv("nmda_modifier")
                g = have[:, GLU]
                if g.any():
                    vg = v[g]
                    t1 = (((inh[g] * rv("Glutamate$g_ampa")[g]).astype(f32) * rv("Glutamate$ampa_r$kinetics$r")[g]).astype(f32)
                          * (vg - rv("Glutamate$e_ampa")[g]).astype(f32)).astype(f32)
                    block = ((numpy_net.expf(((-f32(0.062)) * vg).astype(f32)) * rv("Glutamate$mg")[g]).astype(f32) / f32(3.57)).astype(f32)
                    gate = (f32(1.0) / (f32(1.0) + block).astype(f32)).astype(f32)
                    pw = numpy_net.powf(numpy_net.rust_max(rv("Glutamate$nmda_r$kinetics$r")[g], f32(0.0)), nmod[g])
                    t2 = ((((gate * inh[g]).astype(f32) * rv("Glutamate$g_nmda")[g]).astype(f32) * pw).astype(f32)
                          * (vg - rv("Glutamate$e_nmda")[g]).astype(f32)).astype(f32)
                    rv("Glutamate$current")[g] = (t1 + t2).astype(f32)
                b = have[:, GABA]
                if b.any():
                    rv("GABA$current")[b] = ((rv("GABA$g")[b] * rv("GABA$r$kinetics$r")[b]).astype(f32)
                                             * (v[b] - rv("GABA$e")[b]).astype(f32)).astype(f32)
                dp = have[:, DOPA]
                if dp.any():
                    inh[dp] = (f32(1.0) - (rv("Dopamine$r_d2$kinetics$r")[dp] * rv("Dopamine$s_d2")[dp]).astype(f32)).astype(f32)
                    nmod[dp] = (f32(1.0) - (rv("Dopamine$r_d1$kinetics$r")[dp] * rv("Dopamine$s_d1")[dp]).astype(f32)).astype(f32)
            # on_iteration: du first, then dv, both from the old state, applied after the last statement
            du = (((pa * ((pb * v).astype(f32) - u).astype(f32)).astype(f32) / tau_m).astype(f32) * dt).astype(f32)
            acc = ((((f32(0.04) * v).astype(f32) * v).astype(f32) + (f32(5.0) * v).astype(f32)).astype(f32) + f32(140.0)).astype(f32)
            dv = ((((acc - u).astype(f32) + i_in).astype(f32) / c_m).astype(f32) * dt).astype(f32)
            u_new = (u + du).astype(f32)
            v_new = (v + dv).astype(f32)
            if self.chemical:
                total = np.zeros(self.nn, f32)
                total = np.where(have[:, GLU], (total + rv("Glutamate$current")).astype(f32), total)
                total = np.where(have[:, GABA], (total + rv("GABA$current")).astype(f32), total)
                v_new = (v_new - (total * (dt / c_m).astype(f32)).astype(f32)).astype(f32)
                self.nt_apply("", v_new, prev_spiking, dt)
            spike = v_new >= v_th
            a["current_voltage"] = np.where(spike, pc, v_new).astype(f32)
            C[self.nv["u"]] = np.where(spike, (u_new + pd).astype(f32), u_new).astype(f32)
        a["is_spiking"] = spike.astype(np.uint32)
        a["last_firing_time"] = np.where(spike, np.int32(self.clock), a["last_firing_time"]).astype(np.int32)
        return spike
