#!/usr/bin/env python3
"""bench.py -- neuron-steps/s of the dense 256x256 Izhikevich lattice (BASELINE.json configs[1]).

One "step" = one time-step of the whole lattice (inputs -> neuron update), state and the 17.18 GB
synapse matrix resident in HBM before the timed region.  `--gpus N` > 1 (launched by torchrun, one
rank per GPU) shards the SAME lattice by postsynaptic population (strong scaling) with one RCCL
all-gather of the exchanged planes per step.

Prints ONE JSON line (rank 0) with the driver's contract keys plus `roofline` (dominant kernel
k_inputs_dense timed with HIP events on the stepper's own stream) and `cpu_baseline` (the oracle --
a C restatement of the reference's CPU path -- timed on a bounded column sample, rank 0, N = 1).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ROWS = COLS = 256
MIN_TIMED_S, MAX_REPEATS = 1.0, 4000    # the timed repetitions add up to at least a second (short steps: more repetitions)
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling is ~6.3 TB/s
# HBM traffic of the dominant kernel comes from separate rocprofv3 --pmc passes of THIS command (FETCH_SIZE and
# WRITE_SIZE cannot share a pass); their summaries are committed under profiles/ and quoted here per launch.
def _latest_profile(name):
    """profiles/rNN/<name> of the latest round that holds one"""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]", name)))
    return found[-1] if found else None


PMC_TRAFFIC = {c: _latest_profile(f"{c}_pmc_traffic.json") for c in ("c2", "c3", "c4", "c5", "c6")}


def pmc_traffic(config, world, rows, cols):
    path = PMC_TRAFFIC.get(config)
    if world != 1 or path is None or (rows, cols) != (ROWS, COLS) or not os.path.exists(path):
        return None
    dom = json.load(open(path)).get("dominant_kernel")
    return dom["hbm_traffic_bytes_per_launch"] if dom else None


def pmc_source(config, world, rows, cols):
    """where roofline.traffic comes from: it is NOT measured by this run (PMC counters need their own rocprofv3 passes)"""
    path = PMC_TRAFFIC.get(config)
    if pmc_traffic(config, world, rows, cols) is None:
        return None
    return {"file": os.path.relpath(path, ROOT), "collected": json.load(open(path)).get("collected"),
            "how": "two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of this command, gfx950 corrections of "
                   "MI355X_MICROARCH.md, per launch of the dominant kernel"}


def self_launch(gpus):
    """`python bench.py --gpus N` (N > 1) without a launcher's environment: start the N ranks as a FRESH child --
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same
    arguments>` -- before this process imports torch or makes any GPU call (a process that touched the GPU must never be
    replaced or re-executed on this pool), relay its stdout (rank 0's JSON line is its last line) and return its exit code.
    SNN_BENCH_LAUNCHER replaces the `python -m torch.distributed.run` prefix (tests/test_host_logic.py stubs it)."""
    import shlex
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    prefix = os.environ.get("SNN_BENCH_LAUNCHER")
    cmd = shlex.split(prefix) if prefix else [sys.executable, "-m", "torch.distributed.run"]
    cmd += ["--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
            os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    sys.stdout.write(proc.stdout)
    sys.stdout.flush()
    if proc.returncode != 0:
        print(f"[bench] the launched ranks exited with code {proc.returncode}: {' '.join(cmd)}", file=sys.stderr)
    return proc.returncode


def plasticity_report(dn, pl_ms, pl_steps, spikes_per_step, config):
    """the weight-update launches of a step (spike compaction + STDP scatter), HIP events: time, the bytes the rule has to touch
    and -- dense handles -- the 128-byte LINES those bytes sit in, which is what the scatter moves (its roofline)"""
    if not pl_steps:
        return None
    n_local = sum(e - b for b, e in dn.ranges)
    ms = pl_ms / pl_steps
    useful = 8.0 * (dn.n_tot + n_local) * spikes_per_step
    out = {"ms_per_step": ms, "steps_measured": pl_steps, "touched_bytes_per_step": useful,
           "note": "spike compaction + column/row weight updates of this rank, HIP events; touched bytes = 8 B x (n_tot + n_local) "
                   "per spiking neuron (SURVEY 8d)"}
    if config in ("c2", "c4") and spikes_per_step > 0:
        # quad-row layout: a spiking neuron's column lies in n_tot / 4 lines (16 B of each), its row in n_local / 8 lines
        # (4 B of every fourth word); each line is read and written back
        lines = spikes_per_step * (dn.n_tot / 4.0 + n_local / 8.0)
        line_bytes = lines * 128.0 * 2.0
        out["roofline"] = {"bound": "hbm (line traffic of a scatter)", "line_bytes_per_step": line_bytes,
                           "achieved_line_GBps": line_bytes / (ms * 1e-3) / 1e9, "useful_GBps": useful / (ms * 1e-3) / 1e9,
                           "useful_over_line_bytes": useful / line_bytes, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac_of_peak_useful": useful / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           "frac_of_peak_lines": line_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           "compare_with": "measured_device_ceilings.copy_GBps of the default c2 line (read + write)"}
    return out


def mem_available_bytes():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                return int(line.split()[1]) * 1024
    except OSError:
        pass
    return 16 << 30


def cgroup_cpu_quota():
    """CPUs the container may use at once (cgroup v2 cpu.max / v1 cfs quota), None when unlimited or unknown"""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / p
    except (OSError, ValueError):
        return None


def cpu_baseline(n, threads, budget_s=15.0):
    """Time the oracle (kind "port": a C restatement of the reference's CPU path, dense arrays, all cores = the
    reference's rayon par_iter over postsynaptic neurons, backend/src/neuron/mod.rs:775-790) on the same workload: the
    synaptic-input sums of ALL n postsynaptic neurons when host memory holds the matrix (5 B per synapse: f32 weight +
    u8 connection flag), else of the largest column window that fits in a quarter of MemAvailable.  The matrix is
    streamed in tiles of 1024 columns x 256 rows (snn_o_inputs_tiled: whole 4 KiB row segments per thread, first touched
    by the thread that streams them).  Returns (neuron-steps/s, seconds, steps, sample columns, side measurements)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_binding as ob
    block = 1024
    fit = int(mem_available_bytes() // 4 // (5 * n)) // block * block
    sample_cols = n if fit >= n else max(min(n, block), fit)
    net = ob.Net(n, model=ob.IZHIKEVICH)
    net.arr["weights"] = np.empty((n, sample_cols), np.float32)
    net.arr["connections"] = np.empty((n, sample_cols), np.uint8)
    net.w_col0, net.w_ld = 0, sample_cols
    ob.lib().snn_o_fill_graph_window_blocked(net["weights"].ctypes.data_as(ob.f32p), net["connections"].ctypes.data_as(ob.u8p),
                                             n, n, 0, sample_cols, block, 2, 0.5, 1.5, 0, threads)
    net["gap_conductance"] = 10.0
    net["current_voltage"] = ob.uniform_array(1, n, -65.0, 30.0)
    # The visible CPU count can exceed what the container may actually run at once (a CPU quota): take the thread
    # count that streams a slice of the sample fastest, halving from the visible count.
    probe_cols = min(sample_cols, 8 * block)
    net.n_threads = threads
    net.inputs_tiled(0, probe_cols, block)             # page-in
    best, tried = (0.0, threads), {}
    t = threads
    while t >= 1:
        net.n_threads = t
        t0 = time.perf_counter()
        net.inputs_tiled(0, probe_cols, block)
        rate = probe_cols / (time.perf_counter() - t0)
        tried[t] = rate
        if rate > best[0] * 1.05:
            best = (rate, t)
        if t <= 4:
            break
        t //= 2
    visible, threads = threads, best[1]
    net.n_threads = threads
    t0 = time.perf_counter()
    net.inputs_tiled(0, sample_cols, block)            # warm-up + calibration of the sample length
    one = time.perf_counter() - t0
    steps = int(min(50, max(10 if one < 3.0 else 3, budget_s / max(one, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(steps):
        net.inputs_tiled(0, sample_cols, block)        # O(N) synapses per neuron: the whole per-neuron cost
        net.update_neurons()                           # O(1) per neuron (all N, negligible)
    dt = time.perf_counter() - t0
    extra = {"host_stream_GBps": 5.0 * n * sample_cols * steps / dt / 1e9, "visible_cpus": visible,
             "cgroup_cpu_quota": cgroup_cpu_quota(),
             "thread_count_probe_neuron_steps_per_s": {str(k): v for k, v in tried.items()}}
    # single thread (the reference's parallel = false), on one tile column of the sample
    net.n_threads = 1
    cols1 = min(sample_cols, block)
    t0 = time.perf_counter()
    net.inputs_tiled(0, cols1, block)
    extra["single_thread_value"] = cols1 / (time.perf_counter() - t0)
    extra["all_core_speedup_over_single_thread"] = (sample_cols * steps / dt) / extra["single_thread_value"]
    # BASELINE configs[0] in full: 32x32, 1000 steps, the reference's own CPU-runnable case
    c1 = ob.Net(1024, model=ob.IZHIKEVICH)
    c1["gap_conductance"] = 10.0
    c1["current_voltage"] = ob.uniform_array(1, 1024, -65.0, 30.0)
    c1.fill_graph(2, 0.5, 1.5)
    few = min(threads, 8)               # 1024 neurons = 64 column blocks: more threads only add fork/join cost
    for nthreads, key in ((1, "c1_32x32_1000_steps_single_thread_s"), (few, f"c1_32x32_1000_steps_{few}_threads_s")):
        c1.n_threads = nthreads
        t0 = time.perf_counter()
        c1.run(1000)
        extra[key] = time.perf_counter() - t0
    return sample_cols * steps / dt, dt, steps, sample_cols, threads, extra


def state_checksum(dn, np, dist, rank, world):
    """sha256 over (current_voltage bits, last_firing_time, is_spiking, per-neuron spike totals) of ALL neurons in global
    index order -- each rank contributes the neurons it owns.  Post-population sharding does not change any result
    (the reduction order is defined on the global presynaptic index), so the value must not depend on --gpus."""
    import hashlib
    own = np.asarray(dn.owned, dtype=np.int64)
    cols = []
    for name, dtype in (("current_voltage", np.float32), ("last_firing_time", np.int32), ("is_spiking", np.uint32)):
        full = np.concatenate([dn.get_attr(i, name, dtype=dtype) for i, (_, _, st) in sorted(dn.lattices.items()) if not st])
        cols.append(full[own].view(np.uint32))
    counts = np.concatenate([dn.spike_counts(i) for i, (_, _, st) in sorted(dn.lattices.items()) if not st])
    cols.append(counts[own].astype(np.uint32))
    mine = (own, np.stack(cols))
    parts = [mine]
    if dist is not None:                       # (also at world size 1 under --force-sharded: the same calls every N makes)
        parts = [None] * world
        dist.all_gather_object(parts, mine)
    if rank != 0:
        return None
    n = sum(p[0].size for p in parts)
    table = np.zeros((4, n), np.uint32)
    seen = np.zeros(n, bool)
    for idx, vals in parts:
        table[:, idx] = vals
        seen[idx] = True
    assert seen.all(), "some neuron is owned by no rank"
    return hashlib.sha256(table.tobytes()).hexdigest()


def build_config(args, snn_amd, synthetic, np, rank, world, local_rank):
    """BASELINE.json configs as synthetic inputs.  Returns (handle, neurons, text, kernel)."""
    cfg = args.config
    sharded = world > 1 or args.force_sharded
    fin = (lambda d: d.finalize(rank, world)) if sharded else (lambda d: d.finalize())
    if cfg in ("c1", "c2", "c6"):
        rows, cols = (32, 32) if cfg == "c1" else (args.rows, args.cols)
        n = rows * cols
        dn = snn_amd.DeviceNetwork(model=snn_amd.IZHIKEVICH, device=local_rank)
        dn.add_lattice(0, rows, cols)
        fin(dn)
        # defaults, gap_conductance 10, V0 ~ U[-65,30] seed 1, weights U[0.5,1.5] seed 2, x != y
        dn.set_attr(0, "gap_conductance", np.full(n, 10.0, np.float32))
        dn.set_attr(0, "current_voltage", synthetic.uniform(1, n, -65.0, 30.0))
        dn.fill_graph_synthetic(2, 0.5, 1.5, with_diagonal=False)
        text = (f"{rows}x{cols} Izhikevich lattice, dense gap-junction connectivity (all-to-all, x != y), dt=0.1, "
                f"weights U[0.5,1.5]")
        if cfg == "c6":
            # SURVEY 8f rank 3, not a BASELINE config: the same lattice as a RewardModulatedLattice -- every synapse's
            # weight and trace are rewritten every step (16 B/synapse on top of the 4 B/synapse input pass)
            dn.set_reward_modulator(0, tau_c=0.05, tau_d=5.0, a_plus=0.002, a_minus=0.0015)
            dn.apply_reward(0.01)
            text = "reward-modulated (R-STDP, trace per synapse) " + text
        # 32x32 takes the small-lattice forms: all steps of a run call in one launch (k_run_resident), else one launch per step
        return dn, n, text, ("k_step_resident<0,true,false>" if cfg == "c1" else
                             "k_inputs_rstdp<true,false> (weight + trace rewritten, 16 B per synapse)" if cfg == "c6" else "k_inputs_dense<true,false>")
    if cfg == "c3":
        rows = cols = 128
        n = rows * cols
        dn = snn_amd.DeviceNetwork(model=snn_amd.HODGKIN_HUXLEY, nt_kinetics=snn_amd.NT_DESTEXHE,
                                   receptor_kinetics=snn_amd.RC_DESTEXHE, device=local_rank)
        dn.add_lattice(0, rows, cols)
        fin(dn)
        dn.set_attr(0, "current_voltage", synthetic.uniform(3, n, -70.0, -60.0))
        flags = np.zeros((n, 3), np.uint32)
        flags[:, 0] = 1                                   # AMPA
        dn.set_attr(0, "neurotransmitters$flags", flags)
        dn.set_attr(0, "receptors$flags", flags)
        dn.fill_graph_synthetic(4, 0.5, 1.5, with_diagonal=False)
        dn.set_synapses(True, True)
        return dn, n, ("128x128 Hodgkin-Huxley lattice (Na/K/K-leak gating) + Destexhe AMPA neurotransmitter and "
                       "receptor kinetics, electrical + chemical, dense, dt=0.01"), "k_inputs_dense<true,true>"
    if cfg == "c4":
        n_inh, n_exc = 128 * 128, 256 * 256
        n = n_inh + n_exc
        dn = snn_amd.DeviceNetwork(model=snn_amd.IZHIKEVICH, device=local_rank)
        dn.add_lattice(0, 128, 128)                       # inhibitory pool, id 0
        dn.add_lattice(1, 256, 256)                       # excitatory pool, id 1
        fin(dn)
        for i, m in ((0, n_inh), (1, n_exc)):
            dn.set_attr(i, "gap_conductance", np.full(m, 10.0, np.float32))
        dn.set_attr(0, "current_voltage", synthetic.uniform(4, n_inh, -65.0, 30.0))
        dn.set_attr(1, "current_voltage", synthetic.uniform(4, n_exc, -65.0, 30.0, offset=n_inh))
        # magnitudes U[0.5,1.5]; the sign structure of interacting_pools (inh -> * negative) does not change
        # the traffic and is exercised at test size (tests/test_gpu_network.py)
        dn.fill_graph_synthetic(5, 0.5, 1.5, with_diagonal=False)
        dn.set_plasticity(0)
        dn.set_plasticity(1)
        return dn, n, ("256x256 excitatory + 128x128 inhibitory Izhikevich LatticeNetwork, dense interleaved "
                       "matrix (81 920 neurons), STDP on both lattices, dt=0.1"), "k_inputs_dense<true,false>"
    if cfg == "c5":
        side = args.rows if args.rows != ROWS else 512
        # weak scaling: the lattices grow to (side * world) x side, so that slab r of every lattice -- what rank r owns -- stays
        # side x side: 4 * side^2 neurons (1 M at the default) per rank whatever N is
        rows5 = side * world if args.scaling == "weak" else side
        m = rows5 * side
        dn = snn_amd.DeviceNetwork(model=snn_amd.IZHIKEVICH, spike_train=snn_amd.ST_POISSON, device=local_rank)
        for k in range(4):
            dn.add_lattice(k, rows5, side)
            dn.add_spike_train_lattice(4 + k, rows5, side)
        if sharded:
            # every rank owns the same slab of each of the four lattices: the ring edge k -> k+1 stays inside a rank
            dn.finalize(rank, world, csr=True, by_lattice=True)
        else:
            dn.finalize(csr=True)
        for k in range(4):
            dn.set_attr(k, "gap_conductance", np.full(m, 10.0, np.float32))
            dn.set_attr(k, "current_voltage", synthetic.uniform(6, m, -65.0, 30.0, offset=k * m))
            dn.set_attr(4 + k, "chance_of_firing", np.full(m, 0.01, np.float32))
            dn.set_attr(4 + k, "seed", np.arange(k * m + 1, (k + 1) * m + 1, dtype=np.uint32))   # cell index + 1
        dn.set_graph_csr(*synthetic.c5_csr(rows5, cols=side, posts=dn.owned))
        return dn, 4 * m, (f"4 x ({rows5}x{side}) Izhikevich lattices (radius-2 neighbourhoods) + 4 Poisson spike-train "
                           f"lattices one-to-one + ring k->k+1, CSR, dt=0.1"), "k_step_csr<0,true,false> (inputs + neuron update + Poisson cells in one launch)"
    raise SystemExit(f"unknown --config {cfg}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2", choices=["c1", "c2", "c3", "c4", "c5", "c6"],
                    help="BASELINE.json configs[0..3]; the headline metric is quoted on c2 (default)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=5,
                    help="timed repetitions of --steps steps inside this process; the line reports the median")
    ap.add_argument("--rows", type=int, default=ROWS)
    ap.add_argument("--cols", type=int, default=COLS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--spike-fraction", type=float, default=0.0,
                    help="synthetic drive: before every step this fraction of the neurons is raised above threshold "
                         "(device-side generator, snn_set_synthetic_drive), e.g. --config c4 --spike-fraction 0.001 to "
                         "time STDP under load; 0 (default) leaves the workload as BASELINE.md defines it")
    ap.add_argument("--force-sharded", action="store_true",
                    help="take the multi-GPU code path (shard handle, RCCL exchange per step) even at world size 1")
    ap.add_argument("--stepper", default="library", choices=["library", "torch"],
                    help="who drives a sharded run: snn_run_sharded (the library calls RCCL itself, default) or "
                         "parallel.ShardedStepper (torch.distributed moves the segments)")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="do not bracket k_inputs_dense with HIP events (roofline.achieved is then null)")
    ap.add_argument("--peer-form", action="store_true",
                    help="--config c5, --gpus N > 1: TRY the peer form of the sparse shard step.  The ranks map each other's receive sets "
                         "after the warm-up (IPC handles through torch.distributed) and step in the peer form -- one launch per step, no "
                         "collective (include/snn_amd.h, snn_p2p_*) -- after a trial run; if any rank cannot connect or its trial gives up, "
                         "EVERY rank returns to the collective on a fresh handle (the line says which: peer_form).  Opt-in: the form has "
                         "never run across devices (DESIGN.md section 6); the default is the RCCL halo exchange")
    ap.add_argument("--no-peer-form", action="store_true", help="(the default since round 6; accepted for old command lines)")
    ap.add_argument("--emulate-ranks-on-one-gpu", action="store_true",
                    help="--gpus N > 1 on a box with ONE GPU: every rank is a process on device 0, the process group is gloo and the "
                         "library's collectives go through host memory (parallel.ProcessCollectives).  A rehearsal of the multi-rank "
                         "code path -- agreement, snn_run_sharded, peer-form trial, fall-back, checksums -- NOT a measurement: the line "
                         "says so (transport) and its value means nothing")
    ap.add_argument("--sabotage-peer-trial", type=int, default=-1, help=argparse.SUPPRESS)     # tests: this rank skips its trial run
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="--gpus N > 1: strong = the SAME network sharded N ways (default); weak = the network grows with N so that "
                         "every rank keeps the N = 1 share (c5 only: 4 lattices of 512 N x 512, 1 M neurons per rank)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))
    if args.scaling == "weak" and args.config != "c5":
        raise SystemExit("--scaling weak is defined for --config c5 (a dense lattice N times the size does not fit N times the memory per rank)")

    # dmabuf IPC is the only mode the host driver supports (RCCL across processes); exported by the harness, kept here
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import snn_amd
    from snn_amd import parallel, synthetic

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE)")
    emulated = args.emulate_ranks_on_one_gpu
    if emulated:
        local_rank = 0                                   # every rank on device 0; nothing below may call RCCL
    if local_rank >= torch.cuda.device_count():          # (counting devices does not initialise the GPU)
        raise SystemExit(f"rank {rank}: --gpus {args.gpus} needs device {local_rank}, {torch.cuda.device_count()} visible")
    torch.cuda.set_device(local_rank)
    dist = None
    sharded = world > 1 or args.force_sharded
    red = "cpu" if emulated else "cuda"                  # where the few scalars the ranks reduce live (gloo reduces host tensors)
    if sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if emulated:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    dn, n, workload, kernel_name = build_config(args, snn_amd, synthetic, np, rank, world, local_rank)

    comm = None
    if sharded and args.stepper == "library":
        # the whole step loop inside libsnn_amd.so: kernels -> pack -> RCCL (called by the library on its own second
        # stream) -> unpack -> kernels, ONE host call per run; torch.distributed only carries the 128-byte RCCL id.
        # If ANY rank cannot make its communicator, every rank falls back to the torch.distributed stepper (agreed on
        # through the process group, so that no rank is left alone in a collective).
        try:
            comm = (parallel.ProcessCollectives(dist, rank, world, torch.device("cuda", 0)) if emulated else
                    parallel.LibraryComm(rank, world, local_rank))
            made = 1
        except Exception as e:          # noqa: BLE001
            print(f"[bench] rank {rank}: library communicator failed ({e}); falling back to --stepper torch", file=sys.stderr)
            made = 0
        flag = torch.tensor([made], dtype=torch.int32, device=red)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            if comm is not None:
                comm.close()
            comm = None
            args.stepper = "torch"
    if comm is not None:

        def run(k):
            dn.run_sharded(comm, k)
    elif sharded:
        # torch.distributed moves the segments; everything stream-ordered on torch's current stream: kernels, the
        # RCCL collective, kernels ... no host synchronisation inside the step loop
        side = torch.cuda.Stream()                      # a real (non-default) stream shared by kernels and RCCL ordering
        dn.set_stream(side.cuda_stream)
        stepper = parallel.ShardedStepper(dn, rank, world, always_exchange=True, stream=side,
                                          device=torch.device("cuda", local_rank))

        def run(k):
            stepper.run(k)
            dn.synchronize()
    else:
        run = dn.run

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.spike_fraction > 0:
        dn.set_synthetic_drive(12345, args.spike_fraction, 35.0)
    dn.set_reduced_history(False, False, True)           # per-neuron spike totals (4 B per neuron, in k_update)
    def own_spike_total():
        return sum(int(dn.spike_counts(i).sum()) for i, (_, _, st) in dn.lattices.items() if not st)

    run(args.warmup)
    peer_note, extra_steps = None, 0
    if args.peer_form and args.config == "c5":
        if sharded and world > 1 and comm is not None:
            # parallel.try_peer_form: connect through IPC handles, a trial run, and -- if any rank could not -- every rank back on
            # the collective with a FRESH handle that carries everything this one was given (drive, spike totals, warm-up)
            def rebuild():
                fresh, *_ = build_config(args, snn_amd, synthetic, np, rank, world, local_rank)
                if args.spike_fraction > 0:
                    fresh.set_synthetic_drive(12345, args.spike_fraction, 35.0)
                fresh.set_reduced_history(False, False, True)
                fresh.run_sharded(comm, args.warmup)
                return fresh
            dn, peer_note = parallel.try_peer_form(dn, dist, rank, world, local_rank, run, rebuild,
                                                   sabotage_rank=args.sabotage_peer_trial if args.sabotage_peer_trial >= 0 else None)
        # the step count of the line does not depend on the outcome, nor on N: a completed trial is 8 steps, then a second warm-up
        if peer_note != "taken":
            run(8)
        barrier()
        run(args.warmup)
        extra_steps = 8 + args.warmup
    spikes_before = own_spike_total()
    # HIP events around the dominant kernel: inside the timed region for the streaming configs with ms-scale steps (2 event
    # records per step are noise next to a 2.5 - 13 ms step); for the short-step configs (c1: 4 us, c5: 43 us, c3: 0.19 ms per
    # step) they would cost 2 - 15 % of the step, so those take them in one extra repetition after the timed ones
    events_inline = not args.no_kernel_events and args.config not in ("c1", "c3", "c5")
    events_after = not args.no_kernel_events and not events_inline
    dn.profile_enable(events_inline)
    dn.profile_reset()
    # the timed region (EXACTLY --steps steps between barrier + synchronize on both sides, max over ranks)
    # is repeated --repeats times back to back; the line reports the MEDIAN repetition and lists all of them
    # ... and, for short steps, as many more as it takes for the timed regions to add up to MIN_TIMED_S (the same count on every
    # rank: decided from the max-over-ranks times)
    runs = []
    while len(runs) < max(1, args.repeats) or (sum(runs) < MIN_TIMED_S and len(runs) < MAX_REPEATS):
        barrier()
        t0 = time.perf_counter()
        run(args.steps)
        barrier()
        e = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([e], dtype=torch.float64, device=red)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            e = float(t.item())
        runs.append(e)
        if len(runs) == max(1, args.repeats):
            # N = 1 and N = 8 lines can be diffed: a checksum of the network's state right after the --repeats repetitions asked
            # for (the same number of steps for every N, whatever is added below), every rank's own neurons gathered to rank 0
            state_sha, state_steps = state_checksum(dn, np, dist, rank, world), args.warmup + extra_steps + args.steps * len(runs)
    elapsed = sorted(runs)[len(runs) // 2]
    phases = None
    if events_after:
        spikes_timed = own_spike_total()
        dn.profile_enable(True)
        dn.profile_reset()
        if args.config == "c1" and not sharded:
            dn.set_option("run_timing", 1)              # shader-clock totals of the one-launch run's four phases
        barrier()
        run(args.steps)
        barrier()
        if args.config == "c1" and not sharded and dn.stat("run_timing_steps"):
            clocks = {k: dn.stat("run_timing_" + k) for k in ("poll", "barrier", "turns", "update")}
            phases = {"steps": dn.stat("run_timing_steps"), "clocks_per_step": {k: v / dn.stat("run_timing_steps") for k, v in clocks.items()}}
    launches, kern_ms = dn.profile_read()
    pl_steps, pl_ms = dn.profile_read_plasticity()
    dn.profile_enable(False)
    spikes = (spikes_timed if events_after else own_spike_total()) - spikes_before           # of this rank's own neurons
    if dist is not None:
        t = torch.tensor([spikes], dtype=torch.float64, device=red)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        spikes = int(t.item())
    total_steps = args.steps * len(runs)

    hist_value = None
    if not sharded and args.config == "c2":
        # also report the rate with GridVoltageHistory capture on (one extra 4 B store per neuron-step)
        dn.set_history(voltage=True, spikes=True)
        hsteps = min(args.steps, 100)
        barrier()
        t0 = time.perf_counter()
        run(hsteps)
        barrier()
        hist_value = n * hsteps / (time.perf_counter() - t0)
        dn.set_history(voltage=False, spikes=False)

    spiking_value = None
    if not sharded and args.config == "c2" and args.spike_fraction == 0:
        # the lattice falls silent within the warm-up (gap junctions pull the voltages together): also report the rate with
        # 0.1 % of the population driven above threshold before every step (10 Hz at dt = 0.1 ms), so that the raster ballots,
        # spike totals and last_firing_time stamps are exercised -- AFTER the state checksum, which stays that of the plain run
        dn.set_synthetic_drive(12345, 0.001, 35.0)
        dsteps = min(args.steps, 100)
        run(10)
        before = own_spike_total()
        barrier()
        t0 = time.perf_counter()
        run(dsteps)
        barrier()
        dt_drive = time.perf_counter() - t0
        spiking_value = {"value": n * dsteps / dt_drive, "spikes_per_step": (own_spike_total() - before) / dsteps,
                         "what": "0.1 % of the neurons raised above threshold before every step (snn_set_synthetic_drive)"}
        dn.set_synthetic_drive(12345, 0.0, 35.0)

    # per-rank clocks (first contact with a multi-GPU node: which rank is slow, and how much of a step is the exchange): every
    # rank's own time for --steps steps of the full step, and -- collective form -- of the same loop with a transport that moves
    # nothing (snn_exchange_noop; the state is stale afterwards, which is why this comes after the checksum)
    rank_times = None
    if sharded and world > 1 and comm is not None:
        barrier()
        t0 = time.perf_counter()
        run(args.steps)
        torch.cuda.synchronize()
        full_ms = (time.perf_counter() - t0) / args.steps * 1e3
        only_ms = None
        if peer_note != "taken":
            barrier()
            t0 = time.perf_counter()
            dn.run_sharded_without_exchange(args.steps)
            torch.cuda.synchronize()
            only_ms = (time.perf_counter() - t0) / args.steps * 1e3
        parts = [None] * world
        dist.all_gather_object(parts, (full_ms, only_ms))
        rank_times = {"step_ms_by_rank": [a for a, _ in parts],
                      "compute_only_ms_by_rank": [b for _, b in parts] if only_ms is not None else None,
                      "exchange_ms_by_rank": [a - b for a, b in parts] if only_ms is not None else None}

    ceilings = None
    if rank == 0 and not sharded and args.config == "c2":
        dn.synchronize()
        rd, cp = snn_amd.probe_bandwidth(8 << 30, 5, local_rank)     # this device's own HBM ceilings
        ceilings = {"read_only_GBps": rd, "copy_GBps": cp}

    # evidence of the transport: ncclCommCount of the communicator the library's loop called RCCL on, and what this rank
    # puts on / takes off the wire per step (the exchange plan's segments, 4 B per word)
    rccl_ranks = comm.count()[0] if comm is not None else None
    exchange_bytes = None
    if sharded:
        plan = dn.exchange_plan()
        exchange_bytes = {"mode": plan["mode"], "sent": 4 * int(plan["send_words"]), "received": 4 * int(plan["recv_words"])}
    if rank == 0:
        value = n * args.steps / elapsed
        if not sharded and dn.stat("persistent_run_launches"):
            # all steps of a run call in ONE launch (small electrical-only lattices): `launches` counts its steps
            kernel_name = "k_run_resident<0,true> (many steps per launch; launches = steps)"
        elif dn.stat("steps_sparse_image"):
            # static weights, gap junctions only: the rows read the step image (16-byte records, presynaptic windows in LDS)
            kernel_name = kernel_name.replace("k_step_csr<0,true,false>", "k_step_csr_img<0>")
        elif not sharded and dn.stat("steps_dense_close"):
            # streamed dense matrices: the last workgroup of a column tile also updates the tile's neurons (one launch per step)
            kernel_name = kernel_name.replace("k_inputs_dense<", "k_inputs_dense_close<model,")
        bytes_per_launch = dn.input_kernel_bytes()
        avg_ms = kern_ms / max(1, launches)
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if launches and avg_ms > 0 else 0.0
        out = {
            "metric": "neuron-steps/sec", "value": value, "unit": "neuron-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "repeats": len(runs), "timed_s": sum(runs), "ms_per_step_runs": [r / args.steps * 1e3 for r in runs[:32]],
            "ms_per_step_min": min(runs) / args.steps * 1e3, "ms_per_step_max": max(runs) / args.steps * 1e3,
            "spikes_per_step": spikes / total_steps,
            "state_sha256": state_sha, "state_after_steps": state_steps,
            "plasticity": plasticity_report(dn, pl_ms, pl_steps, spikes / total_steps / world, args.config),
            # strong: --gpus N shards the SAME network (total work fixed as N grows); weak (c5): every rank keeps the N = 1 share
            "scaling": args.scaling,
            "stepper": (("library (snn_run_sharded, RCCL called by libsnn_amd.so)" if comm is not None else
                         "torch (parallel.ShardedStepper, torch.distributed moves the segments)") if sharded else None),
            "rccl_ranks": None if emulated else rccl_ranks, "exchange_bytes_per_rank_step": exchange_bytes,
            "transport": ("EMULATION: every rank a process on ONE GPU, collectives through host memory and gloo -- not a measurement"
                          if emulated else ("RCCL" if sharded else None)),
            "halo_peer_steps": (dn.stat("halo_peer_steps") if sharded else None), "peer_form": peer_note, "rank_times": rank_times,
            "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "neurons": n,
                       "parallelism": f"post-population shards x{world}" if world > 1 else "single GPU"},
            "value_with_voltage_and_spike_history": hist_value,
            "value_with_0p1pct_of_the_neurons_spiking_per_step": spiking_value,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc_traffic(args.config, world, args.rows, args.cols),
                         "traffic_source": pmc_source(args.config, world, args.rows, args.cols),
                         "kernel": kernel_name, "measured_device_ceilings": ceilings,
                         "frac_of_measured_read_ceiling": (achieved / ceilings["read_only_GBps"]) if ceilings else None,
                         "launches": launches, "avg_launch_ms": avg_ms,
                         "kernel_events": ("inside the timed region" if events_inline else
                                           "one extra repetition after the timed ones" if events_after else "off"),
                         "algorithmic_bytes_per_launch": bytes_per_launch},
        }
        if world == 1 and not args.no_cpu_baseline and args.config in ("c1", "c2"):
            visible_cpus = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            v, secs, cpu_steps, sample, threads, extra = cpu_baseline(n, visible_cpus)
            quota = cgroup_cpu_quota()
            cores = int(min(visible_cpus, quota)) if quota else visible_cpus
            # cores = what the container may run at once (CPU affinity capped by the cgroup quota); threads = the OpenMP team
            # the thread-count probe picked (more threads than cores can still stream faster: they hide memory latency)
            out["cpu_baseline"] = {"value": v, "unit": "neuron-steps/s", "cores": max(1, cores), "threads": threads, "kind": "port", **extra,
                                   "sample": f"oracle (C restatement, OpenMP x{threads}, tiles of 1024 columns x 256 rows) on "
                                             f"{sample} of {n} postsynaptic neurons x {cpu_steps} steps ({secs:.1f} s); every "
                                             f"sampled neuron sums all {n} presynaptic terms, i.e. the full per-neuron-step "
                                             f"cost; bound by host memory bandwidth (host_stream_GBps, 5 B per synapse)"}
        if phases is not None:
            # the one-launch run of a small lattice reads its matrix from HBM once per RUN: no HBM fraction describes it.
            # What bounds a step is the chain poll -> barrier -> the four serial turns of the canonical sum -> update and
            # publish; reported as measured shares of the launch (workgroup 0's shader clocks, scaled to the event time)
            per_step_us = avg_ms * 1e3
            tot = sum(phases["clocks_per_step"].values())
            out["roofline"] = {"bound": "latency", "achieved": None, "peak": None, "unit": "us/step", "frac": None, "traffic": None,
                               "kernel": kernel_name, "launches": launches, "avg_step_us": per_step_us,
                               "phases_us_per_step": {k: per_step_us * v / tot for k, v in phases["clocks_per_step"].items()},
                               "phases_clocks_per_step": phases["clocks_per_step"],
                               "matrix_bytes_read_once_per_run": bytes_per_launch,
                               "kernel_events": "one extra repetition after the timed ones",
                               "note": "all steps of a run call in one launch; the matrix stays in registers / LDS, per step "
                                       "only 8 B per neuron travel (L2 / fabric)"}
        result = json.dumps(out)
    else:
        result = None

    dn.close()
    if comm is not None:
        comm.close()
    if dist is not None:
        dist.destroy_process_group()
    if result is not None:
        # RCCL writes its version banner through C stdio, which a pipe buffers until exit: flush it first so
        # that the JSON line is the LAST line on stdout
        import ctypes
        ctypes.CDLL(None).fflush(None)
        print(result, flush=True)


if __name__ == "__main__":
    main()
