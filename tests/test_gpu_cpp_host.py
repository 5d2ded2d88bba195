"""The C++ host mirror (host/snn_lattice.hpp: Lattice, LatticeGPU::from_lattice/run_lattice,
LatticeNetworkGPU::from_network/run_lattices) driven from a compiled C++ program, checked against the oracle."""
import os
import subprocess

import numpy as np
import pytest

import oracle_binding as ob
import parity

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def v_init(i, j):
    return np.float32(-65.0) + np.float32(5.0) * np.float32((i * 7 + j * 3) % 19)


def weight(a, b):
    return np.float32(0.5) + np.float32(0.0625) * np.float32((a[0] + 2 * a[1] + 3 * b[0] + 5 * b[1]) % 16)


def test_cpp_host_lattice_and_network(tmp_path, snn):
    from snn_amd import _lib
    exe = tmp_path / "host_lattice_test"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.run(["g++", "-std=c++17", "-O1", "-o", str(exe), os.path.join(ROOT, "tests", "cpp", "host_lattice_test.cpp"),
                    "-L" + libdir, "-lsnn_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    r = subprocess.run([str(exe), str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr

    # ---- single lattice ----
    rows, cols = 6, 7
    lay = parity.Layout([(0, rows, cols)])
    net = parity.make_oracle(lay)
    net["gap_conductance"] = 10.0
    pos = [(i, j) for i in range(rows) for j in range(cols)]
    net["current_voltage"] = np.array([v_init(i, j) for i, j in pos], np.float32)
    for a, pa in enumerate(pos):
        for b, pb in enumerate(pos):
            if pa != pb:
                net["connections"][a, b] = 1
                net["weights"][a, b] = weight(pa, pb)
    net.run(250, voltage_history=True)
    hist = np.fromfile(tmp_path / "lattice_history.f32", np.float32).reshape(250, rows * cols)
    assert np.array_equal(hist.view(np.uint32), net.voltage_history.view(np.uint32))
    fin = np.fromfile(tmp_path / "lattice_final.f32", np.float32).reshape(rows * cols, 3)
    assert np.array_equal(fin[:, 0].view(np.uint32), net["current_voltage"].view(np.uint32))
    assert np.array_equal(fin[:, 1].view(np.uint32), net["w_value"].view(np.uint32))
    assert np.array_equal(fin[:, 2].astype(np.int32), net["last_firing_time"])

    # ---- network ----
    lay = parity.Layout([(1, 4, 4)], [(0, 4, 4)])
    net = parity.make_oracle(lay, st_kind=ob.ST_RATE)
    net["gap_conductance"] = 10.0
    pos = [(i, j) for i in range(4) for j in range(4)]
    net["current_voltage"] = np.array([v_init(j, i) for i, j in pos], np.float32)
    for a, pa in enumerate(pos):
        for b, pb in enumerate(pos):
            if pa != pb:
                net["connections"][a, b] = 1
                net["weights"][a, b] = weight(pa, pb)
        net["connections"][16 + a, a] = 1
        net["weights"][16 + a, a] = 2.0
    net["st_rate"] = 3.0
    net["do_plasticity"] = 1
    net.run(400, summaries=True, spike_counts=True)
    avg = np.fromfile(tmp_path / "network_average_voltage.f32", np.float32)
    eeg = np.fromfile(tmp_path / "network_eeg.f32", np.float32)
    assert np.array_equal(avg.view(np.uint32), net.avg_history[:, 0].view(np.uint32))
    assert np.array_equal(eeg.view(np.uint32), net.eeg_history[:, 0].view(np.uint32))
    counts = np.fromfile(tmp_path / "network_spike_counts.f32", np.float32)
    assert np.array_equal(counts.astype(np.uint32), net.spike_counts)
    w = np.fromfile(tmp_path / "network_weights.f32", np.float32).reshape(16, 16)
    ow = np.where(net["connections"][:16] != 0, net["weights"][:16], np.float32(np.nan))
    assert np.array_equal(np.isnan(w), np.isnan(ow))
    assert np.array_equal(np.nan_to_num(w).view(np.uint32), np.nan_to_num(ow).view(np.uint32))
    cw = np.fromfile(tmp_path / "network_connecting_weights.f32", np.float32)
    assert np.array_equal(cw.view(np.uint32), net["weights"][16:][np.arange(16), np.arange(16)].view(np.uint32))
    vf = np.fromfile(tmp_path / "network_final_v.f32", np.float32)
    assert np.array_equal(vf.view(np.uint32), net["current_voltage"].view(np.uint32))

    # ---- reward-modulated lattice through the C++ mirror ----
    lay = parity.Layout([(0, 3, 3)])
    net = parity.make_oracle(lay)
    net["gap_conductance"] = 10.0
    pos = [(i, j) for i in range(3) for j in range(3)]
    net["current_voltage"] = np.array([v_init(j, i) for i, j in pos], np.float32)
    for a, pa in enumerate(pos):
        for b, pb in enumerate(pos):
            if pa != pb:
                net["connections"][a, b] = 1
                net["weights"][a, b] = weight(pa, pb)
    net["rm_do_modulation"] = 1
    net["rm_tau_c"] = 0.05
    net["rm_a_plus"] = 0.01
    net["rm_a_minus"] = 0.01
    rewards = np.array([0.5 if t % 50 == 0 else 0.0 for t in range(600)], np.float32)
    net.run(600, rewards=rewards)
    rw = np.fromfile(tmp_path / "reward_weights.f32", np.float32).reshape(9, 9)
    ow = np.where(net["connections"] != 0, net["weights"], np.float32(np.nan))
    assert np.array_equal(np.isnan(rw), np.isnan(ow))
    assert np.array_equal(np.nan_to_num(rw).view(np.uint32), np.nan_to_num(ow).view(np.uint32))
    rt = np.fromfile(tmp_path / "reward_traces.f32", np.float32).reshape(9, 9)
    assert np.array_equal(rt.view(np.uint32), net["traces"].view(np.uint32))
    assert np.fromfile(tmp_path / "reward_dopamine.f32", np.float32)[0] == net["rm_dopamine"][0] != 0
