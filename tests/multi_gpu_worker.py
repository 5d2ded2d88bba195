"""One rank of a real multi-GPU run (launched by tests/test_gpu_multi_device.py through torch.distributed.run, one
process per GPU): its shard handle of a small network, the library's own step loop snn_run_sharded over RCCL, the state
of the neurons it owns written to <out>/rank<r>.npz for the parent to hold against the oracle.  Test infrastructure."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def build(form):
    """the networks of the check: a dense two-lattice network with chemical synapses, Rate cells and STDP; the structure of
    BASELINE configs[4] (sparse, contiguous shards or shards by lattice)"""
    import numpy as np
    import oracle_binding as ob
    import parity
    if form.startswith("csr"):
        from test_gpu_csr import c5_structure
        net = c5_structure(32)                       # 4 x 32 x 32 neurons + as many Poisson cells: 8 shards of 512
        net["do_plasticity"] = 1
        return net
    lay = parity.Layout([(0, 24, 24), (3, 16, 32)], [(5, 4, 8)])            # 1088 neurons (five chunks), 32 Rate cells
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
    net["stdp_a_plus"][1] = 1.5
    return net


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--form", required=True, choices=["dense", "csr", "csr_by_lattice"])
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank, world, local = (int(os.environ[k]) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"))
    import numpy as np
    import torch
    import torch.distributed as dist
    import snn_amd
    from snn_amd import parallel
    import parity
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    net = build(args.form)
    csr = args.form.startswith("csr")
    dn = parity.device_from_oracle(snn_amd, net, shard=(rank, world), device=local, csr=csr,
                                   by_lattice=(args.form == "csr_by_lattice"))
    comm = parallel.LibraryComm(rank, world, local)
    dn.run_sharded(comm, args.steps // 2)             # sparse handles trade their halo lists inside the first call
    dn.run_sharded(comm, args.steps - args.steps // 2)
    st = parity.pull_state(dn, net)
    plan = dn.exchange_plan()
    own = np.zeros(net.n_neurons, bool)
    own[dn.owned] = True
    known = own.copy()
    if plan["mode"] == "halo":
        for p in range(world):
            if p != rank:
                known[dn.halo_needs(p)] = True
    else:
        known[:] = True
    w = dn.get_graph_csr() if csr else dn.get_graph_rows(0, net.n_tot)[0]
    np.savez(os.path.join(args.out, f"rank{rank}.npz"), own=own, known=known, mode=plan["mode"], weights=w,
             clock=dn.clock, device=torch.cuda.current_device(),
             **{k: st[k] for k in ("current_voltage", "is_spiking", "last_firing_time", "w_value", "nt_t")})
    dn.close()
    comm.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
