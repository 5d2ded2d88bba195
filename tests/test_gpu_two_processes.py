"""TWO processes, each holding one shard handle on the SAME GPU (RCCL refuses two ranks on one device -- "Duplicate GPU
detected" -- so the bytes travel through host memory and gloo): the library's own sharded step loop
(snn_run_sharded_custom) with the product's plan, pack and unpack kernels on both sides of a real process boundary.
Dense (all-gather of voltage / transmitter plane / spike bits) and sparse (halo segments, lists traded between the
ranks) against the single-process oracle."""
import os
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS = 300


def build(form):
    import oracle_binding as ob
    import parity
    if form.startswith("csr"):
        from test_gpu_csr import c5_structure
        net = c5_structure(16 if form.startswith("csr_by_lattice") else 8)
        net["do_plasticity"] = 0 if form.endswith("_peer") else 1       # (the peer form is the step without weight updates)
        return net
    lay = parity.Layout([(0, 9, 10), (3, 12, 12)], [(5, 3, 4)])
    net = parity.make_oracle(lay, st_kind=ob.ST_RATE, chemical=True)
    nn, nc = net.n_neurons, net.n_cells
    net["current_voltage"] = ob.uniform_array(1, nn, -65.0, 30.0)
    net["gap_conductance"] = 10.0
    net["nt_flags"][:, 0] = 1
    net["rc_flags"][:, 0] = 1
    net["rc_g"][:, 0] = 2.0
    net["st_nt_flags"][:, 0] = 1
    net["st_rate"] = ob.uniform_array(4, nc, 1.0, 5.0)
    net.fill_graph(2, 0.5, 1.5)
    rng = np.random.default_rng(3)
    net["connections"][rng.random(net["connections"].shape) < 0.3] = 0
    net["do_plasticity"] = 1
    return net


def worker(rank, world, init_file, out_dir, form):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import snn_amd
    from snn_amd import parallel
    import parity
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    net = build(form)
    dn = parity.device_from_oracle(snn_amd, net, shard=(rank, world), csr=form.startswith("csr"),
                                   by_lattice=form.startswith("csr_by_lattice"))
    if form.startswith("csr"):
        # the need lists cross the process boundary too: what I read of peer p becomes p's send list for me
        needs = [dn.halo_needs(p) if p != rank else np.zeros(0, np.uint32) for p in range(world)]
        gathered = [None] * world
        dist.all_gather_object(gathered, needs)
        for p in range(world):
            if p != rank:
                dn.halo_set_sends(p, gathered[p][rank])
        dn.halo_commit()
    plan = dn.exchange_plan()
    dev = torch.device("cuda", 0)
    if form.endswith("_peer"):
        # the PEER form across a real process boundary: IPC handles of the receive sets and done counters travel (gloo), every
        # rank maps its neighbours' and the border rows store straight into them -- no transport function at all
        busy = int(plan["send_words"]) + int(plan["recv_words"]) > 0          # (a rank that owns nothing exchanges nothing)
        mine = None
        if busy:
            loc = dn.p2p_local()
            mine = (dn.p2p_ipc_export(), [int(x) for x in loc["offsets"]], [int(x) for x in loc["counts"]])
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        for p in range(world):
            if busy and p != rank and everyone[p] is not None and (everyone[p][2][rank] or mine[2][p]):
                r0, r1, fl = dn.p2p_ipc_import(everyone[p][0])
                dn.p2p_connect(p, r0, r1, fl, everyone[p][1][rank])
        if busy:
            dn.p2p_commit()
        dist.barrier()
        dn.run_sharded_without_exchange(STEPS // 2)
        dn.run_sharded_without_exchange(STEPS - STEPS // 2)
        assert dn.stat("halo_peer_steps") == (STEPS if busy else 0)
        exchange = None
    send, recv = parallel.exchange_tensors(plan, dev)

    def exchange(_stream):
        dn_stream_sync()
        if plan["mode"] == "allgather":
            block = send.numel()
            parts = [torch.empty(block, dtype=torch.int32) for _ in range(world)]
            dist.all_gather(parts, send.cpu())
            for p in range(world):
                if p != rank:
                    recv[p * block:(p + 1) * block].copy_(parts[p])
        else:
            ins = [int(x) for x in plan["send_count"]]
            outs = [int(x) for x in plan["recv_count"]]
            got = torch.empty(sum(outs), dtype=torch.int32)
            dist.all_to_all_single(got, send.cpu(), output_split_sizes=outs, input_split_sizes=ins)
            recv.copy_(got)
        torch.cuda.synchronize()

    def dn_stream_sync():
        torch.cuda.synchronize()            # the handle's stream is a blocking-free stream of this device: wait for all

    if not form.endswith("_peer"):
        dn.run_sharded_custom(exchange, STEPS // 2)
        dn.run_sharded_custom(exchange, STEPS - STEPS // 2)
    st = parity.pull_state(dn, net)
    known = np.zeros(net.n_neurons, bool)
    own = np.zeros(net.n_neurons, bool)
    own[dn.owned] = True
    known |= own
    if plan["mode"] == "halo":
        for p in range(world):
            if p != rank:
                known[dn.halo_needs(p)] = True
    else:
        known[:] = True
    w = dn.get_graph_csr() if form.startswith("csr") else dn.get_graph_rows(0, net.n_tot)[0]
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), own=own, known=known, mode=plan["mode"], recv_words=plan["recv_words"],
             weights=w, clock=dn.clock, **{k: st[k] for k in ("current_voltage", "is_spiking", "last_firing_time", "w_value", "nt_t")})
    dn.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("form,world", [("dense", 2), ("csr", 2), ("csr", 3), ("csr_by_lattice", 2), ("csr_peer", 2), ("csr_peer", 3),
                                        ("csr_by_lattice_peer", 2), ("csr_by_lattice_peer", 4)])
def test_library_step_loop_across_processes_on_one_gpu(snn, form, world):
    import torch.multiprocessing as mp
    import parity
    with tempfile.TemporaryDirectory() as d:
        mp.get_context("spawn")
        mp.spawn(worker, args=(world, os.path.join(d, "rendezvous"), d, form), nprocs=world, join=True)
        ref = build(form)
        ref.run(STEPS, spike_history=True)
        assert ref.spike_history.sum() > 20
        covered = np.zeros(ref.n_neurons, bool)
        for r in range(world):
            z = np.load(os.path.join(d, f"rank{r}.npz"))
            own, k = z["own"], z["known"]
            assert str(z["mode"]) == ("halo" if form.startswith("csr") else "allgather") and int(z["clock"]) == STEPS
            if form in ("csr", "csr_peer") and world > 2:
                assert not k.all()                     # a genuinely partial view
            if form.startswith("csr_by_lattice"):      # only the slab borders travel: 2 lattice rows x 16 x 4 lattices
                assert int(z["recv_words"]) <= (2 if world > 2 else 1) * (4 * 2 * 16 + 8)       # (two neighbours from 3 ranks up)
            covered |= own
            for name in ("current_voltage", "is_spiking", "last_firing_time"):
                assert np.array_equal(parity.bits(z[name][k]), parity.bits(ref[name][k])), (r, name)
            assert np.array_equal(parity.bits(z["nt_t"][own]), parity.bits(ref["nt_t"][own]))
            assert np.array_equal(parity.bits(z["w_value"][own]), parity.bits(ref["w_value"][own]))
            if form.startswith("csr"):
                _, _, want = parity.csr_for_posts(ref, np.flatnonzero(own))
                assert np.array_equal(parity.bits(z["weights"]), parity.bits(want))
            else:
                ow = np.where(ref["connections"] != 0, ref["weights"], np.float32(0))
                assert np.array_equal(parity.bits(z["weights"][:, own]), parity.bits(ow[:, own]))
        assert covered.all()
