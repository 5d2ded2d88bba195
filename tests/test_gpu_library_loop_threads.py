"""The library's OWN sharded loop (snn_run_sharded: kernels, pack, the collective on a second stream, the own-rows input pass
or the interior slices overlapping it, unpack; the ranks' agreement and the trading of halo lists before the first step) with
MORE THAN ONE rank on one GPU: RCCL refuses two ranks on a device, so the collectives are replaced process-wide
(snn_set_collectives) by parallel.ThreadCollectives -- every rank a host thread, all-gather and grouped send / receive as
device-to-device copies between the ranks' buffers.  Everything but RCCL's own kernels is the code a multi-GPU run executes.
Against the oracle."""
import threading

import numpy as np
import pytest

import parity
from test_gpu_csr import c5_structure
from test_gpu_sharded import build as dense_net

pytestmark = [pytest.mark.gpu, pytest.mark.emulated_ranks]


def run_ranks(handles, tc, calls):
    errors = []

    def work(r):
        try:
            for steps in calls:
                handles[r].run_sharded(tc.comm(r), steps)
        except BaseException as e:       # noqa: BLE001
            errors.append(e)
            tc.abort()

    # daemon threads, and a barrier that is broken when the waiting main thread is interrupted (pytest-timeout): a stuck rank
    # must neither outlive the test nor keep the interpreter from exiting
    threads = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(len(handles))]
    for t in threads:
        t.start()
    try:
        for t in threads:
            t.join()
    except BaseException:
        tc.abort()
        raise
    if errors:
        raise errors[0]


@pytest.fixture
def collectives(snn):
    import torch
    from snn_amd import parallel
    made = []

    def make(world):
        tc = parallel.ThreadCollectives(world, torch.device("cuda", 0))
        made.append(tc)
        return tc
    yield make
    for tc in made:
        tc.close()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("n_shards,chemical,plastic", [(2, False, False), (3, True, False), (4, False, True), (3, True, True)])
def test_dense_shards_all_gather_in_the_library_loop(snn, collectives, n_shards, chemical, plastic):
    """without weight updates the next step's own-rows input pass is enqueued before the loop waits for the collective"""
    net = dense_net(chemical)
    net["do_plasticity"] = int(plastic)
    handles = [parity.device_from_oracle(snn, net, shard=(r, n_shards)) for r in range(n_shards)]
    tc = collectives(n_shards)
    run_ranks(handles, tc, [200, 1, 119])
    assert tc.calls["all_gather"] == 320 + 2            # one per step + the two words of the agreement before the first run
    net.run(320, spike_history=True)
    assert net.spike_history.sum() > (20 if plastic else -1)        # (the static network stays below threshold: traces only)
    for h in handles:
        assert h.clock == 320
        st = parity.pull_state(h, net)
        parity.assert_shard_view_equal(h, st, net)
        b, e = h.post_begin, h.post_end
        w, _ = h.get_graph_rows(0, net.n_tot)
        ow = np.where(net["connections"] != 0, net["weights"], np.float32(0))
        assert np.array_equal(parity.bits(w[:, b:e]), parity.bits(ow[:, b:e]))
        for name in ("w_value", "nt_t", "rc_r"):
            assert np.array_equal(parity.bits(st[name][b:e]), parity.bits(net[name][b:e])), name
        h.close()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("n_shards,by_lattice,plastic", [(2, True, False), (4, True, False), (3, False, False), (8, False, False),
                                                         (4, True, True), (2, False, True)])
def test_sparse_shards_trade_lists_and_halos_in_the_library_loop(snn, collectives, n_shards, by_lattice, plastic):
    """the halo lists are not wired by the test: the ranks' agreement finds them uncommitted and trades them (grouped send /
    receive); without weight updates the rows then read the received segments directly, two launches per step"""
    net = c5_structure(16 if by_lattice else 8)
    net["do_plasticity"] = int(plastic)
    handles = [parity.device_from_oracle(snn, net, shard=(r, n_shards), csr=True, by_lattice=by_lattice) for r in range(n_shards)]
    tc = collectives(n_shards)
    run_ranks(handles, tc, [150, 50])
    net.run(200, spike_history=True)
    assert net.spike_history.sum() > (20 if by_lattice else 5)
    for r, h in enumerate(handles):
        plan = h.exchange_plan()
        assert plan["mode"] == "halo" and h.clock == 200
        if h.owned.size and int(plan["send_words"]) + int(plan["recv_words"]):
            assert h.stat("halo_direct_steps") == (0 if plastic else 200)
        st = parity.pull_state(h, net)
        parity.assert_shard_view_equal(h, st, net)
        cells = h.cells_read()
        for name in ("st_last_firing_time", "st_seed"):
            assert np.array_equal(parity.bits(st[name][cells]), parity.bits(net[name][cells])), name
        assert np.array_equal(parity.bits(st["w_value"][h.owned]), parity.bits(net["w_value"][h.owned]))
        parity.assert_graph_equal(net, h)
        h.close()


def test_replaced_collectives_are_restored(snn, collectives):
    """after close() the library calls RCCL again: a communicator of world size 1 made by the library itself works"""
    from snn_amd import parallel
    tc = collectives(2)
    tc.close()
    comm = parallel.LibraryComm(0, 1, 0)
    net = dense_net(False)
    dn = parity.device_from_oracle(snn, net, shard=(0, 1))
    dn.run_sharded(comm, 20)
    net.run(20)
    st = parity.pull_state(dn, net)
    assert np.array_equal(parity.bits(st["current_voltage"]), parity.bits(net["current_voltage"]))
    dn.close()
    comm.close()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("n_shards,by_lattice", [(4, True), (5, False), (7, False)])
def test_larger_sparse_network_in_the_library_loop(snn, collectives, n_shards, by_lattice):
    """4 x (32 x 32) neurons + Poisson cells: every rank has several workgroups of rows (dealt to the XCDs in bands), border and
    interior slices, a halo of more than one segment; shard counts that do not divide the population"""
    net = c5_structure(32)
    handles = [parity.device_from_oracle(snn, net, shard=(r, n_shards), csr=True, by_lattice=by_lattice) for r in range(n_shards)]
    tc = collectives(n_shards)
    run_ranks(handles, tc, [210, 90])
    net.n_threads = 8
    net.run(300, spike_history=True)
    assert net.spike_history.sum() > 20
    for h in handles:
        assert h.clock == 300
        if h.owned.size:
            assert h.stat("halo_direct_steps") == 300
        st = parity.pull_state(h, net)
        parity.assert_shard_view_equal(h, st, net)
        cells = h.cells_read()
        assert np.array_equal(parity.bits(st["st_seed"][cells]), parity.bits(net["st_seed"][cells]))
        assert np.array_equal(parity.bits(st["w_value"][h.owned]), parity.bits(net["w_value"][h.owned]))
        h.close()
