"""snn_run_sharded -- the sharded step loop inside the library with RCCL called directly -- on the one GPU a test box
has: (1) a compiled C++ host that uses nothing but include/snn_amd.h (the library makes the communicator, world size
1) against the oracle; (2) the same entry point from Python next to the torch-driven ShardedStepper."""
import os
import subprocess

import numpy as np
import pytest

import oracle_binding as ob
import parity

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("form", ["dense", "csr"])
def test_cpp_host_runs_a_shard_through_the_c_abi(tmp_path, snn, form):
    from snn_amd import _lib
    exe = tmp_path / "run_sharded_test"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run(["g++", "-std=c++17", "-O1", "-o", str(exe), os.path.join(ROOT, "tests", "cpp", "run_sharded_test.cpp"),
                    "-L" + libdir, "-lsnn_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    steps = 300
    env = dict(os.environ, NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1")
    r = subprocess.run([str(exe), str(tmp_path), form, str(steps)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr

    rows, cols = 9, 11
    n = rows * cols
    net = parity.make_oracle(parity.Layout([(0, rows, cols)]))
    net["gap_conductance"] = 10.0
    i = np.arange(n)
    net["current_voltage"] = (np.float32(-65.0) + np.float32(6.0) * ((i * 7) % 19).astype(np.float32)).astype(np.float32)
    p, q = np.meshgrid(i, i, indexing="ij")
    conn = (p != q) & ((p * 31 + q * 17) % 5 != 0)
    net["connections"][...] = conn
    net["weights"][...] = np.where(conn, np.float32(0.5) + np.float32(0.0625) * ((p * 3 + q * 5) % 16).astype(np.float32), 0)
    net["do_plasticity"] = 1
    net.run(steps + 10, spike_history=True)
    assert net.spike_history.sum() > 5
    v = np.fromfile(tmp_path / "v.f32", np.float32)
    w = np.fromfile(tmp_path / "w_value.f32", np.float32)
    lft = np.fromfile(tmp_path / "lft.i32", np.int32)
    assert np.array_equal(v.view(np.uint32), net["current_voltage"].view(np.uint32))
    assert np.array_equal(w.view(np.uint32), net["w_value"].view(np.uint32))
    assert np.array_equal(lft, net["last_firing_time"])


@pytest.mark.timeout(600)
def test_run_sharded_from_python_with_a_library_made_communicator(snn):
    from snn_amd import parallel
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    os.environ.setdefault("NCCL_IB_DISABLE", "1")
    lay = parity.Layout([(0, 12, 12), (2, 7, 9)], [(5, 3, 4)])
    net = parity.make_oracle(lay, st_kind=ob.ST_POISSON, chemical=True)
    nn, nc = net.n_neurons, net.n_cells
    net["gap_conductance"] = 10.0
    net["current_voltage"] = ob.uniform_array(1, nn, -65.0, 30.0)
    net["nt_flags"][:, 2] = 1
    net["rc_flags"][:, 2] = 1
    net["st_nt_flags"][:, 0] = 1
    net["st_chance_of_firing"] = 0.03
    net.fill_graph(2, 0.5, 1.5)
    dn = parity.device_from_oracle(snn, net, shard=(0, 1))
    comm = parallel.LibraryComm(0, 1, 0)
    plan = dn.exchange_plan()
    # voltage + the one transmitter type a NEURON releases (GABA); the cells' AMPA never travels (cells are replicated)
    assert plan["mode"] == "allgather" and plan["plane_id"] == [0, 4]
    dn.run_sharded(comm, 150)
    dn.set_synapses(True, False)                 # the plan follows the synapse kinds: voltage + spike bits only
    net.electrical, net.chemical = True, False
    net.run(150)
    assert dn.exchange_plan()["plane_id"] == [0]
    dn.run_sharded(comm, 120)
    net.run(120)
    parity.assert_state_equal(net, parity.pull_state(dn, net))
    parity.assert_graph_equal(net, dn)
    dn.close()
    comm.close()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("form", ["dense", "csr", "csr_by_lattice"])
def test_the_ranks_agreement_on_the_exchange_runs_at_world_size_one(snn, form, monkeypatch):
    """Before the first step of snn_run_sharded the ranks compare their plans over the communicator (graph form, halo
    lists committed or not, planes on the wire) and sparse handles trade their need lists.  With a single GPU that code
    is skipped (one rank has nobody to disagree with); SNN_AMD_ALWAYS_AGREE walks every collective of it once -- an
    in-place ncclAllGather of one word per rank, the list exchange, the plan comparison -- over a real RCCL communicator."""
    from snn_amd import parallel
    from test_gpu_csr import c5_structure
    monkeypatch.setenv("SNN_AMD_ALWAYS_AGREE", "1")
    os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
    os.environ.setdefault("NCCL_IB_DISABLE", "1")
    if form == "dense":
        net = parity.make_oracle(parity.Layout([(0, 9, 9)]))
        net["gap_conductance"] = 10.0
        net["current_voltage"] = ob.uniform_array(1, net.n_neurons, -65.0, 30.0)
        net.fill_graph(2, 0.5, 1.5)
    else:
        net = c5_structure(8)
    net["do_plasticity"] = 1
    dn = parity.device_from_oracle(snn, net, shard=(0, 1), csr=(form != "dense"), by_lattice=(form == "csr_by_lattice"))
    comm = parallel.LibraryComm(0, 1, 0)
    dn.run_sharded(comm, 200)
    dn.run_sharded(comm, 100)                     # agreed: no second round
    net.run(300)
    parity.assert_state_equal(net, parity.pull_state(dn, net))
    parity.assert_graph_equal(net, dn)
    dn.close()
    comm.close()
