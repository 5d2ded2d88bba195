"""No allocation failure -- device memory, page-locked memory or a host-side table -- leaves the C ABI as anything but a status
code (the reference returns Result<_, GPUError>, error/mod.rs:221-238; include/snn_amd.h: "never abort").  The failure hook
(snn_debug_fail_alloc_at) makes the n-th allocation of a create / finalize / set / run / get sequence fail, for EVERY n the
sequence makes: each outcome is an SnnError with a message (or, where the library has a fall-back for that allocation, the
unchanged result), the handle can be destroyed, and the next handle of the process computes what the oracle computes."""
import ctypes as C

import numpy as np
import pytest

import oracle_binding as ob
import parity

pytestmark = pytest.mark.gpu


def arm(snn, n):
    seen = C.c_uint64()
    snn._lib.check(snn._lib.load().snn_debug_fail_alloc_at(int(n), C.byref(seen)))
    return int(seen.value)


CASES = {
    # dense, STDP, Rate cells, both kinds of synapse, histories: the two-kernel / one-launch step, the one-launch run
    "dense": dict(layout=([(0, 6, 7), (1, 3, 3)], [(5, 2, 3)]), csr=False, chemical=True, plastic=True, steps=12),
    # sparse handle: SELL slices, gather plan, transpose index for plasticity
    "sparse": dict(layout=([(0, 9, 9)], [(3, 3, 3)]), csr=True, chemical=False, plastic=True, steps=9),
    # more than 1024 rows: the streamed step forms and a run that takes the one-launch run of larger networks
    "larger": dict(layout=([(0, 34, 34)], []), csr=False, chemical=False, plastic=False, steps=6),
}


def make_oracle(case, seed):
    lat, st = case["layout"]
    net = parity.make_oracle(parity.Layout(lat, st), model=ob.IZHIKEVICH, st_kind=ob.ST_RATE if st else ob.ST_NONE, chemical=case["chemical"])
    net["gap_conductance"] = 10.0
    net["current_voltage"] = ob.uniform_array(seed, net.n_neurons, -65.0, 30.0)
    net.fill_graph(seed + 1, 0.1, 0.6)
    if st:
        net["st_rate"] = 0.3 + 0.1 * (np.arange(net.n_cells) % 4)
    if case["chemical"]:
        net["nt_flags"][:, 0] = 1
        net["rc_flags"][:, 0] = 1
        net["st_nt_flags"][:, 0] = 1
    if case["plastic"]:
        net["do_plasticity"][...] = 1
    return net


def whole_sequence(snn, case, net):
    """create ... destroy through the Python mirror; returns what the getters read"""
    dn = parity.device_from_oracle(snn, net, csr=case["csr"])
    try:
        dn.set_history(voltage=True, spikes=True)
        dn.run(case["steps"])
        dn.run(3)                                             # a second call: history regrow, plans already built
        out = {"v": dn.get_attr(0, "current_voltage"), "lft": dn.get_attr(0, "last_firing_time", np.int32),
               "vh": dn.voltage_history(0), "sh": dn.spike_history(0), "clock": dn.clock}
        if case["csr"]:
            out["w"] = dn.get_graph_csr()
        else:
            out["w"], out["c"] = dn.get_graph_rows(0, net.n_tot)
        return out
    finally:
        dn.close()


@pytest.mark.parametrize("name", sorted(CASES))
def test_every_allocation_of_a_sequence_may_fail(snn, name):
    case = CASES[name]
    net = make_oracle(case, 11)
    try:
        before = arm(snn, 0)
        want = whole_sequence(snn, case, net)
        total = arm(snn, 0) - before
        assert total > 20, f"only {total} allocations counted: is the hook wired to the allocators?"
        outcomes = {"error": 0, "unchanged": 0}
        messages = set()
        for n in range(1, total + 1):
            arm(snn, n)
            try:
                got = whole_sequence(snn, case, net)
            except snn.SnnError as e:
                assert e.code != 0 and str(e).split(":", 1)[1].strip(), "a failed allocation must come back as a code with a message"
                assert e.code in (3, 4, 5, 6, 8, 12), f"allocation {n} of {total}: unexpected status {e.code}: {e}"
                outcomes["error"] += 1
                messages.add(str(e)[:120])
                continue
            finally:
                arm(snn, 0)
            # the library had a fall-back for this allocation (or it was one of a path the armed pass did not take): same results
            for k in want:
                assert np.array_equal(np.atleast_1d(got[k]).view(np.uint8), np.atleast_1d(want[k]).view(np.uint8)), (n, k)
            outcomes["unchanged"] += 1
        assert outcomes["error"] >= total * 0.8, (outcomes, total)
        assert any("bad_alloc" in m for m in messages), "no host-side table among the failed allocations?"
        assert any("bad_alloc" not in m for m in messages), "no device allocation among the failed ones?"
        # after total failures the process still computes the oracle's results
        got = whole_sequence(snn, case, net)
        for k in want:
            assert np.array_equal(np.atleast_1d(got[k]).view(np.uint8), np.atleast_1d(want[k]).view(np.uint8)), k
        onet = make_oracle(case, 11)
        onet.run(case["steps"] + 3, voltage_history=True)
        assert np.array_equal(onet["current_voltage"][:want["v"].size].view(np.uint32), want["v"].view(np.uint32))      # (lattice 0 = the first neurons)
    finally:
        arm(snn, 0)


def test_a_handle_that_failed_half_way_is_still_a_handle(snn):
    """after a failed finalize / run the handle answers further calls with status codes and can be destroyed"""
    case = CASES["dense"]
    net = make_oracle(case, 5)
    try:
        for n in (1, 3, 9, 27, 60):
            dn = snn.DeviceNetwork(model=net.model, spike_train=net.st_kind)
            for i, r, c in net.layout.lattices:
                dn.add_lattice(i, r, c)
            for i, r, c in net.layout.st_lattices:
                dn.add_spike_train_lattice(i, r, c)
            arm(snn, n)
            codes = []
            for call in (dn.finalize, lambda: dn.set_synapses(True, False), lambda: dn.run(5), lambda: dn.get_attr(0, "current_voltage"),
                         lambda: dn.run(2), lambda: dn.set_history(voltage=True, spikes=False), lambda: dn.run(2)):
                try:
                    call()
                    codes.append(0)
                except snn.SnnError as e:
                    codes.append(e.code)
            arm(snn, 0)
            assert any(codes), (n, codes)
            dn.close()
    finally:
        arm(snn, 0)
