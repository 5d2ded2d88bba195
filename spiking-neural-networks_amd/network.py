"""`DeviceNetwork`: thin numpy-facing owner of one C-ABI handle (include/snn_amd.h).

Mirrors the reference's GPU containers at buffer level -- `LatticeGPU` /
`LatticeNetworkGPU` (backend/src/neuron/gpu_lattices/mod.rs:327-350, 1517-1558):
named per-cell buffers (`IterateAndSpikeGPU::convert_to_gpu`,
neuron/iterate_and_spike/mod.rs:3156-3189), the flattened graph
(`InterleavingGraphGPU`, graph/mod.rs:579-943) and `run_lattices`.
The Lixirnet-style classes in `lattice.py` sit on top of this.
"""
import ctypes as C
import os

import numpy as np

from . import _lib

IZHIKEVICH, LIF, HODGKIN_HUXLEY, QUADRATIC_INTEGRATE_AND_FIRE, SIMPLE_LIF = 0, 1, 2, 3, 4
ADAPTIVE_LIF, ADAPTIVE_EXP_LIF, LEAKY_IZHIKEVICH, BCM_IZHIKEVICH = 5, 6, 7, 8
CUSTOM = 100          # the generated model of a library built by _lib.build_custom (modelgen.py)
NT_APPROXIMATE, NT_DESTEXHE, NT_DISCRETE_SPIKE, NT_EXPONENTIAL_DECAY = 0, 1, 2, 3
RC_APPROXIMATE, RC_DESTEXHE, RC_EXPONENTIAL_DECAY = 0, 1, 2
ST_NONE, ST_POISSON, ST_RATE, ST_PRESET, ST_BCM_POISSON = 0, 1, 2, 3, 4
ST_CUSTOM = 100       # the generated spike train of a library built by _lib.build_custom (modelgen.parse_description)
NT_CUSTOM = RC_CUSTOM = 100   # generated [neurotransmitter_kinetics] / [receptor_kinetics] of such a library
REFRACTORINESS_CUSTOM = 2   # neural_refractoriness$kind of the generated refractoriness of such a library
NUM_NT_TYPES = 3

_DT = {np.dtype(np.float32): "f32", np.dtype(np.uint32): "u32", np.dtype(np.int32): "i32"}
_POISON = os.environ.get("SNN_HOST_POISON", "")


def _out(shape, dtype=np.float32):
    """A buffer the library is about to fill.  SNN_HOST_POISON=1 (campaigns): pre-filled with a pattern no result holds --
    a signalling-NaN-like word for floats, 0xDB bytes otherwise -- so that a getter which returns before its transfer has
    landed, or fills less than it reports, shows in every comparison instead of depending on what the allocator handed out."""
    a = np.empty(shape, dtype)
    if _POISON:
        if a.dtype == np.float32:
            a.view(np.uint32)[...] = 0x7FA0DEAD
        else:
            a.view(np.uint8)[...] = 0xDB
    return a
_PTR = {"f32": _lib.f32p, "u32": _lib.u32p, "i32": _lib.i32p}


class DeviceNetwork:
    def __init__(self, model=IZHIKEVICH, nt_kinetics=NT_APPROXIMATE, receptor_kinetics=RC_APPROXIMATE,
                 spike_train=ST_NONE, device=0, lib_path=None):
        self._L = _lib.load(lib_path)
        self._h = _lib.H()
        self._check(self._L.snn_network_create(device, model, nt_kinetics, receptor_kinetics, spike_train,
                                              C.byref(self._h)))
        self.model = model
        self.lattices = {}          # id -> (rows, cols, is_spike_train)
        self.finalized = False

    def _check(self, code):
        _lib.check(code, self._L)

    # ---- lifetime -------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.snn_network_destroy(self._h)
            self._h = _lib.H()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- construction ---------------------------------------------------------------------
    def add_lattice(self, id, rows, cols):
        self._check(self._L.snn_network_add_lattice(self._h, id, rows, cols))
        self.lattices[id] = (rows, cols, False)

    def add_spike_train_lattice(self, id, rows, cols):
        self._check(self._L.snn_network_add_spike_train_lattice(self._h, id, rows, cols))
        self.lattices[id] = (rows, cols, True)

    def finalize(self, shard_index=None, n_shards=None, csr=False, by_lattice=False):
        """csr=True: the handle holds a sparse CSR graph (set_graph_csr) instead of a dense matrix; by_lattice=True
        (sparse shard handles): the shard owns slab `shard_index` of EVERY neuron lattice instead of one contiguous slot"""
        if csr:
            self._check(self._L.snn_network_use_csr(self._h, 1))
        self.csr = bool(csr)
        if shard_index is None:
            self._check(self._L.snn_network_finalize(self._h))
        elif by_lattice:
            self._check(self._L.snn_network_finalize_shard_by_lattice(self._h, shard_index, n_shards))
        else:
            self._check(self._L.snn_network_finalize_shard(self._h, shard_index, n_shards))
        self.finalized = True
        nn, nc, q0, q1 = (C.c_uint32() for _ in range(4))
        self._check(self._L.snn_network_sizes(self._h, C.byref(nn), C.byref(nc), C.byref(q0), C.byref(q1)))
        self.n_neurons, self.n_cells = nn.value, nc.value
        self.n_tot = self.n_neurons + self.n_cells
        self.post_begin, self.post_end = q0.value, q1.value
        n = C.c_uint32()
        self._check(self._L.snn_shard_ranges(self._h, None, None, 0, C.byref(n)))
        b, e = np.zeros(n.value, np.uint32), np.zeros(n.value, np.uint32)
        if n.value:
            self._check(self._L.snn_shard_ranges(self._h, b.ctypes.data_as(_lib.u32p), e.ctypes.data_as(_lib.u32p), n.value, C.byref(n)))
        self.ranges = [(int(x), int(y)) for x, y in zip(b, e)]          # owned [begin, end), ascending
        return self

    @property
    def owned(self):
        """global indices of the neurons this handle owns, ascending (the row order of set_graph_csr)"""
        return (np.concatenate([np.arange(b, e, dtype=np.int64) for b, e in self.ranges]) if self.ranges
                else np.zeros(0, np.int64))

    def lattice_range(self, id):
        first, count = C.c_uint32(), C.c_uint32()
        self._check(self._L.snn_network_lattice_range(self._h, id, C.byref(first), C.byref(count)))
        return first.value, count.value

    # ---- attributes -----------------------------------------------------------------------
    def set_attr(self, id, name, values):
        a = np.ascontiguousarray(values)
        if a.dtype not in _DT:
            raise TypeError(f"attribute arrays must be float32 / uint32 / int32, not {a.dtype}")
        kind = _DT[a.dtype]
        fn = getattr(self._L, f"snn_set_attr_{kind}")
        self._check(fn(self._h, id, name.encode(), a.ctypes.data_as(_PTR[kind]), a.size))

    def get_attr(self, id, name, dtype=np.float32, per_type=False):
        rows, cols, _ = self.lattices[id]
        n = rows * cols
        out = _out((n, NUM_NT_TYPES) if per_type else (n,), dtype=dtype)
        kind = _DT[np.dtype(dtype)]
        fn = getattr(self._L, f"snn_get_attr_{kind}")
        self._check(fn(self._h, id, name.encode(), out.ctypes.data_as(_PTR[kind]), out.size))
        return out

    # ---- graph ----------------------------------------------------------------------------
    def set_graph_dense(self, weights, connections):
        w = np.ascontiguousarray(weights, dtype=np.float32)
        c = np.ascontiguousarray(connections, dtype=np.uint32)
        if w.shape != c.shape or w.ndim != 2 or w.shape[0] != w.shape[1]:
            raise ValueError("weights / connections must be equal square matrices [n_tot, n_tot]")
        self._check(self._L.snn_set_graph_dense(self._h, w.ctypes.data_as(_lib.f32p), c.ctypes.data_as(_lib.u32p),
                                               w.shape[0]))

    def get_graph_dense(self):
        n = self.n_tot
        w = np.zeros((n, n), np.float32)
        c = np.zeros((n, n), np.uint32)
        self._check(self._L.snn_get_graph_dense(self._h, w.ctypes.data_as(_lib.f32p), c.ctypes.data_as(_lib.u32p), n))
        return w, c

    def set_graph_rows(self, pre_begin, weights, connections):
        """rows [pre_begin, pre_begin+len) of the [n_tot, n_neurons] matrix"""
        w = np.ascontiguousarray(weights, dtype=np.float32)
        c = np.ascontiguousarray(connections, dtype=np.uint32)
        if w.shape != c.shape or w.ndim != 2 or w.shape[1] != self.n_neurons:
            raise ValueError("row blocks must be [rows, n_neurons]")
        self._check(self._L.snn_set_graph_rows(self._h, pre_begin, w.shape[0], w.ctypes.data_as(_lib.f32p),
                                              c.ctypes.data_as(_lib.u32p)))

    def get_graph_rows(self, pre_begin, pre_count):
        w = np.zeros((pre_count, self.n_neurons), np.float32)
        c = np.zeros((pre_count, self.n_neurons), np.uint32)
        self._check(self._L.snn_get_graph_rows(self._h, pre_begin, pre_count, w.ctypes.data_as(_lib.f32p),
                                              c.ctypes.data_as(_lib.u32p)))
        return w, c

    def set_graph_csr(self, row_ptr, pre_index, weights):
        """CSR by OWNED postsynaptic neuron in ascending global order (self.owned): row_ptr[n_owned + 1], pre_index
        ascending inside each row"""
        rp = np.ascontiguousarray(row_ptr, dtype=np.uint64)
        pi = np.ascontiguousarray(pre_index, dtype=np.uint32)
        w = np.ascontiguousarray(weights, dtype=np.float32)
        if rp.size != sum(e - b for b, e in self.ranges) + 1 or pi.size != w.size:
            raise ValueError("row_ptr must have one entry per owned neuron + 1 and pre_index / weights equal lengths")
        self._nnz = int(w.size)
        self._check(self._L.snn_set_graph_csr(self._h, rp.ctypes.data_as(_lib.u64p), pi.ctypes.data_as(_lib.u32p),
                                             w.ctypes.data_as(_lib.f32p), w.size))

    def get_graph_csr(self):
        w = _out(getattr(self, "_nnz", 0), np.float32)
        self._check(self._L.snn_get_graph_csr(self._h, w.ctypes.data_as(_lib.f32p), w.size))
        return w

    def fill_graph_synthetic(self, seed, lo, hi, with_diagonal=False):
        self._check(self._L.snn_fill_graph_synthetic(self._h, seed, lo, hi, int(with_diagonal)))

    # ---- switches -------------------------------------------------------------------------
    def set_synapses(self, electrical=True, chemical=False):
        self._check(self._L.snn_set_synapses(self._h, int(electrical), int(chemical)))

    def set_plasticity(self, id, a_plus=2.0, a_minus=2.0, tau_plus=4.5, tau_minus=4.5, dt=0.1, do_plasticity=True):
        self._check(self._L.snn_set_plasticity(self._h, id, a_plus, a_minus, tau_plus, tau_minus, dt,
                                              int(do_plasticity)))

    def set_history(self, voltage=False, spikes=False):
        self._check(self._L.snn_set_history(self._h, int(voltage), int(spikes)))

    def reset_history(self):
        self._check(self._L.snn_reset_history(self._h))

    def reset_timing(self):
        self._check(self._L.snn_reset_timing(self._h))

    @property
    def clock(self):
        v = C.c_uint64()
        self._check(self._L.snn_get_clock(self._h, C.byref(v)))
        return v.value

    def set_clock(self, clock):
        """internal_clock of the network (LatticeNetworkGPU::from_network): firing times are absolute step numbers"""
        self._check(self._L.snn_set_clock(self._h, int(clock)))

    def set_spike_train_clock(self, id, clock):
        self._check(self._L.snn_set_spike_train_clock(self._h, id, int(clock)))

    def spike_train_clock(self, id):
        v = C.c_uint64()
        self._check(self._L.snn_get_spike_train_clock(self._h, id, C.byref(v)))
        return v.value

    # ---- test support ---------------------------------------------------------------------
    def checkpoint(self):
        """keeps everything a later run call reads (snn_debug_checkpoint); restore_checkpoint puts it back"""
        self._check(self._L.snn_debug_checkpoint(self._h, 0))

    def restore_checkpoint(self):
        self._check(self._L.snn_debug_checkpoint(self._h, 1))

    def verify_report(self):
        """what the last mismatch of the option "verify" was ("" when there was none)"""
        return (self._L.snn_debug_verify_report(self._h) or b"").decode()

    # ---- stepping -------------------------------------------------------------------------
    def run(self, iterations):
        self._check(self._L.snn_run(self._h, iterations))

    def step_begin(self):
        self._check(self._L.snn_step_begin(self._h))

    def step_end(self):
        self._check(self._L.snn_step_end(self._h))

    def step_begin_local(self):
        self._check(self._L.snn_step_begin_local(self._h))

    def refresh_begin(self):
        """packs the current state of the plan's planes when the mirror lacks one of them (snn_refresh_begin); True if so"""
        needed = C.c_int(0)
        self._check(self._L.snn_refresh_begin(self._h, C.byref(needed)))
        return bool(needed.value)

    def refresh_end(self):
        self._check(self._L.snn_refresh_end(self._h))

    def exchange_plan(self):
        """The shard handle's exchange plan (snn_exchange_plan + per-peer segments): a dict with mode ("allgather" |
        "halo"), n_shards, shard_index, shard_stride, planes, plane_id, send / recv (device pointers), send_words /
        recv_words, and per-peer arrays send_offset / send_count / recv_offset / recv_count in 32-bit words."""
        plan = _lib.ExchangePlan()
        self._check(self._L.snn_exchange_plan_get(self._h, C.byref(plan)))
        g = plan.n_shards
        arrs = [np.zeros(g, np.uint64) for _ in range(4)]
        self._check(self._L.snn_exchange_peers(self._h, *[a.ctypes.data_as(_lib.u64p) for a in arrs]))
        return {"mode": "halo" if plan.mode == _lib.EXCHANGE_HALO else "allgather", "n_shards": g,
                "shard_index": plan.shard_index, "shard_stride": plan.shard_stride, "planes": plan.planes,
                "plane_id": list(plan.plane_id)[:plan.planes], "send": plan.send, "recv": plan.recv,
                "send_words": plan.send_words, "recv_words": plan.recv_words,
                "send_offset": arrs[0], "send_count": arrs[1], "recv_offset": arrs[2], "recv_count": arrs[3]}

    def p2p_local(self):
        """the peer form's endpoints of this handle (snn_p2p_local): recv0, recv1, flags addresses + per-shard offsets / counts"""
        r0, r1, fl = C.c_uint64(), C.c_uint64(), C.c_uint64()
        g = self.exchange_plan()["n_shards"]
        off, cnt = np.zeros(g, np.uint64), np.zeros(g, np.uint64)
        self._check(self._L.snn_p2p_local(self._h, C.byref(r0), C.byref(r1), C.byref(fl), off.ctypes.data_as(_lib.u64p),
                                          cnt.ctypes.data_as(_lib.u64p)))
        return {"recv": (r0.value, r1.value), "flags": fl.value, "offsets": off, "counts": cnt}

    def p2p_ipc_export(self):
        """192 bytes: the IPC handles of the two receive sets and the done counters (snn_p2p_ipc_export), for a peer in ANOTHER process"""
        buf = C.create_string_buffer(192)
        self._check(self._L.snn_p2p_ipc_export(self._h, buf))
        return bytes(buf.raw)

    def p2p_ipc_import(self, handles, device=0):
        """maps a peer's exported handles into this process: (recv0, recv1, flags) addresses for p2p_connect"""
        r0, r1, fl = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._check(self._L.snn_p2p_ipc_import(device, C.create_string_buffer(handles, 192), C.byref(r0), C.byref(r1), C.byref(fl)))
        return r0.value, r1.value, fl.value

    def p2p_ipc_close(self, recv0, recv1, flags, device=0):
        """unmaps what p2p_ipc_import mapped (after the peers have stopped stepping in the peer form)"""
        self._check(self._L.snn_p2p_ipc_close(device, int(recv0), int(recv1), int(flags)))

    def p2p_connect(self, peer, recv0, recv1, flags, recv_offset):
        self._check(self._L.snn_p2p_connect(self._h, peer, int(recv0), int(recv1), int(flags), int(recv_offset)))

    def p2p_commit(self):
        self._check(self._L.snn_p2p_commit(self._h))

    def halo_needs(self, peer):
        """ascending global indices of shard `peer`'s neurons that this handle's CSR rows read"""
        n = C.c_uint32()
        self._check(self._L.snn_halo_needs(self._h, peer, None, 0, C.byref(n)))
        out = _out(n.value, np.uint32)
        if n.value:
            self._check(self._L.snn_halo_needs(self._h, peer, out.ctypes.data_as(_lib.u32p), out.size, C.byref(n)))
        return out

    def cells_read(self):
        """indices (into the cell population) of the spike-train cells this handle advances: all of them, or -- sparse
        shard handles of a multi-rank run -- the ones its own rows read"""
        n = C.c_uint32()
        self._check(self._L.snn_cells_read(self._h, None, 0, C.byref(n)))
        out = _out(n.value, np.uint32)
        if n.value:
            self._check(self._L.snn_cells_read(self._h, out.ctypes.data_as(_lib.u32p), out.size, C.byref(n)))
        return out

    def halo_set_sends(self, peer, indices):
        a = np.ascontiguousarray(indices, dtype=np.uint32)
        self._check(self._L.snn_halo_set_sends(self._h, peer, a.ctypes.data_as(_lib.u32p), a.size))

    def halo_commit(self):
        self._check(self._L.snn_halo_commit(self._h))

    def exchange(self, comm):
        """one RCCL exchange of the packed segments on the handle's stream (comm: parallel.LibraryComm or ncclComm_t)"""
        self._check(self._L.snn_exchange(self._h, C.c_void_p(int(comm))))

    def exchange_halo_lists(self, comm):
        self._check(self._L.snn_comm_exchange_halo_lists(self._h, C.c_void_p(int(comm))))

    def run_sharded_without_exchange(self, iterations):
        """the library's sharded step loop with snn_exchange_noop as the transport (timing of one rank's step)"""
        fn = C.cast(self._L.snn_exchange_noop, C.c_void_p)
        self._check(self._L.snn_run_sharded_custom(self._h, fn, None, int(iterations)))

    def run_sharded_custom(self, exchange, iterations):
        """the library's sharded step loop with the caller's transport: `exchange(hip_stream)` is called once per step
        after the outgoing segments were enqueued and must have moved the plan's segments when it returns (with the option
        "halo_direct" 2 it reads the pointers of `exchange_plan()` anew at every call: two sets of segments alternate)"""
        err = []

        def thunk(_user, stream):
            try:
                exchange(stream)
                return 0
            except BaseException as e:      # never let an exception cross the C frame
                err.append(e)
                return 1

        cb = _lib.EXCHANGE_FN(thunk)
        try:
            self._check(self._L.snn_run_sharded_custom(self._h, C.cast(cb, C.c_void_p), None, int(iterations)))
        except Exception:
            if err:
                raise err[0]
            raise

    def run_sharded(self, comm, iterations):
        """`iterations` steps with the library's own RCCL loop (every rank calls it with its shard handle)"""
        self._check(self._L.snn_run_sharded(self._h, C.c_void_p(int(comm)), int(iterations)))

    def set_stream(self, hip_stream):
        """Adopt a caller's hipStream_t (int handle): step_begin / step_end then only enqueue on it.
        None returns to the handle's own stream.  The legacy default stream (handle 0) cannot be adopted --
        create a real stream (e.g. torch.cuda.Stream()) so that collectives can be ordered against it."""
        if hip_stream is None:
            self._check(self._L.snn_set_stream(self._h, None))
            return
        if int(hip_stream) == 0:
            raise ValueError("the default (null) stream cannot be adopted: pass a non-default stream handle")
        self._check(self._L.snn_set_stream(self._h, C.c_void_p(int(hip_stream))))

    def synchronize(self):
        self._check(self._L.snn_synchronize(self._h))

    def stream(self):
        p = C.c_void_p()
        self._check(self._L.snn_stream(self._h, C.byref(p)))
        return p.value

    # ---- histories ------------------------------------------------------------------------
    def history_steps(self):
        v = C.c_uint64()
        self._check(self._L.snn_history_steps(self._h, C.byref(v)))
        return v.value

    def voltage_history(self, id):
        rows, cols, _ = self.lattices[id]
        steps = self.history_steps()
        out = _out((steps, rows * cols), np.float32)
        self._check(self._L.snn_get_voltage_history(self._h, id, out.ctypes.data_as(_lib.f32p), out.size))
        return out

    def spike_history(self, id):
        rows, cols, _ = self.lattices[id]
        steps = self.history_steps()
        out = _out((steps, rows * cols), np.uint8)
        self._check(self._L.snn_get_spike_history(self._h, id, out.ctypes.data_as(_lib.u8p), out.size))
        return out

    def set_bcm(self, id, decay=0.1, average_scalar=0.1, dt=0.1, do_plasticity=True):
        """BCM rule (plasticity/mod.rs:72-116) for lattice `id` instead of STDP"""
        self._check(self._L.snn_set_bcm(self._h, id, decay, average_scalar, dt, int(do_plasticity)))

    # ---- reward modulation (RewardModulatedLattice, neuron/mod.rs:2719-3417) -----------------
    def set_reward_modulator(self, id, dopamine=0.0, tau_d=20.0, tau_c=0.0001, a_plus=2.0, a_minus=2.0, tau_plus=4.5,
                             tau_minus=4.5, dt=0.1, do_modulation=True):
        self._check(self._L.snn_set_reward_modulator(self._h, id, dopamine, tau_d, tau_c, a_plus, a_minus, tau_plus,
                                                    tau_minus, dt, int(do_modulation)))

    def dopamine(self, id):
        out = C.c_float()
        self._check(self._L.snn_get_dopamine(self._h, id, C.byref(out)))
        return np.float32(out.value)

    def apply_reward(self, reward):
        self._check(self._L.snn_apply_reward(self._h, reward))

    def run_with_reward(self, reward):
        self._check(self._L.snn_run_with_reward(self._h, reward))

    def set_trace_rows(self, pre_begin, traces):
        t = np.ascontiguousarray(traces, dtype=np.float32)
        if t.ndim != 2 or t.shape[1] != self.n_neurons:
            raise ValueError("traces must be [rows][n_neurons]")
        self._check(self._L.snn_set_trace_rows(self._h, pre_begin, t.shape[0], t.ctypes.data_as(_lib.f32p)))

    def get_trace_rows(self, pre_begin, pre_count):
        t = np.zeros((pre_count, self.n_neurons), np.float32)
        self._check(self._L.snn_get_trace_rows(self._h, pre_begin, pre_count, t.ctypes.data_as(_lib.f32p)))
        return t

    # ---- connections between lattices of a reward-modulated network (include/snn_amd.h, snn_set_connection_kind) ----
    def set_connection_kind(self, pre_id, post_id, kind):
        """0 a plain network's connection, 1 RewardModulatedConnection::RewardModulatedWeight, 2 RewardModulatedConnection::Weight"""
        self._check(self._L.snn_set_connection_kind(self._h, pre_id, post_id, int(kind)))

    def set_pending_rows(self, pre_begin, pending):
        t = np.ascontiguousarray(pending, dtype=np.float32)
        if t.ndim != 2 or t.shape[1] != self.n_neurons:
            raise ValueError("pending must be [rows][n_neurons]")
        self._check(self._L.snn_set_pending_rows(self._h, pre_begin, t.shape[0], t.ctypes.data_as(_lib.f32p)))

    def get_pending_rows(self, pre_begin, pre_count):
        t = np.zeros((pre_count, self.n_neurons), np.float32)
        self._check(self._L.snn_get_pending_rows(self._h, pre_begin, pre_count, t.ctypes.data_as(_lib.f32p)))
        return t

    def set_counter_rows(self, pre_begin, counters):
        """TraceRSTDP::counter (0 / 1) per connection, rows as set_pending_rows"""
        t = np.ascontiguousarray(counters, dtype=np.uint8)
        if t.ndim != 2 or t.shape[1] != self.n_neurons:
            raise ValueError("counters must be [rows][n_neurons]")
        self._check(self._L.snn_set_counter_rows(self._h, pre_begin, t.shape[0], t.ctypes.data_as(_lib.u8p)))

    def get_counter_rows(self, pre_begin, pre_count):
        t = np.zeros((pre_count, self.n_neurons), np.uint8)
        self._check(self._L.snn_get_counter_rows(self._h, pre_begin, pre_count, t.ctypes.data_as(_lib.u8p)))
        return t

    def set_traces_csr(self, traces):
        t = np.ascontiguousarray(traces, dtype=np.float32)
        self._check(self._L.snn_set_traces_csr(self._h, t.ctypes.data_as(_lib.f32p), t.size))

    def get_traces_csr(self):
        t = _out(getattr(self, "_nnz", 0), np.float32)
        self._check(self._L.snn_get_traces_csr(self._h, t.ctypes.data_as(_lib.f32p), t.size))
        return t

    def set_pending_csr(self, pending):
        """TraceRSTDP::dw per stored edge of a sparse handle, in the edge order of set_graph_csr"""
        t = np.ascontiguousarray(pending, dtype=np.float32)
        self._check(self._L.snn_set_pending_csr(self._h, t.ctypes.data_as(_lib.f32p), t.size))

    def get_pending_csr(self):
        t = _out(getattr(self, "_nnz", 0), np.float32)
        self._check(self._L.snn_get_pending_csr(self._h, t.ctypes.data_as(_lib.f32p), t.size))
        return t

    def set_counters_csr(self, counters):
        t = np.ascontiguousarray(counters, dtype=np.uint8)
        self._check(self._L.snn_set_counters_csr(self._h, t.ctypes.data_as(_lib.u8p), t.size))

    def get_counters_csr(self):
        t = _out(getattr(self, "_nnz", 0), np.uint8)
        self._check(self._L.snn_get_counters_csr(self._h, t.ctypes.data_as(_lib.u8p), t.size))
        return t

    def set_firing_times(self, id, cell_ptr, times):
        """PresetSpikeTrain firing times of spike-train lattice `id`: cell i fires through
        times[cell_ptr[i]:cell_ptr[i+1]] cyclically (spike_train/mod.rs:753-833)."""
        cp = np.ascontiguousarray(cell_ptr, dtype=np.uint32)
        t = np.ascontiguousarray(times, dtype=np.float32)
        rows, cols, _ = self.lattices[id]
        if cp.size != rows * cols + 1:
            raise ValueError("cell_ptr must have rows*cols + 1 entries")
        self._check(self._L.snn_set_firing_times(self._h, id, cp.ctypes.data_as(_lib.u32p),
                                                t.ctypes.data_as(_lib.f32p), t.size))

    def set_graph_history(self, id, enable=True):
        """enable: 0 off, 1 (True) snapshot after the step's weight updates (a lone Lattice), 2 before them (the order of
        LatticeNetwork::iterate)"""
        self._check(self._L.snn_set_graph_history(self._h, id, int(enable)))

    def graph_history(self, id):
        """[steps][n][n] snapshots of lattice `id`'s internal weights (update_graph_history)"""
        rows, cols, _ = self.lattices[id]
        n = rows * cols
        out = _out((self.history_steps(), n, n), np.float32)
        self._check(self._L.snn_get_graph_history(self._h, id, out.ctypes.data_as(_lib.f32p), out.shape[0]))
        return out

    def set_history_stride(self, every):
        self._check(self._L.snn_set_history_stride(self._h, int(every)))

    def set_reduced_history(self, average_voltage=False, eeg=False, spike_counts=False,
                            reference_voltage=0.007, distance=0.8, conductivity=251.0):
        self._check(self._L.snn_set_reduced_history(self._h, int(average_voltage), int(eeg), int(spike_counts),
                                                   reference_voltage, distance, conductivity))

    def average_voltage_history(self, id):
        out = _out(self.history_steps(), np.float32)
        self._check(self._L.snn_get_average_voltage_history(self._h, id, out.ctypes.data_as(_lib.f32p), out.size))
        return out

    def eeg_history(self, id):
        out = _out(self.history_steps(), np.float32)
        self._check(self._L.snn_get_eeg_history(self._h, id, out.ctypes.data_as(_lib.f32p), out.size))
        return out

    def spike_counts(self, id):
        rows, cols, _ = self.lattices[id]
        out = _out(rows * cols, np.uint32)
        self._check(self._L.snn_get_spike_counts(self._h, id, out.ctypes.data_as(_lib.u32p), out.size))
        return out

    def set_option(self, name, value):
        """tuning switches of include/snn_amd.h (fused_step, defer_rstdp, defer_stdp, uniform_params, persistent_run, input_shape)"""
        self._check(self._L.snn_set_option(self._h, name.encode(), int(value)))

    def stat(self, name):
        """launch counters of include/snn_amd.h (persistent_run_launches, persistent_run_steps)"""
        out = C.c_uint64(0)
        self._check(self._L.snn_get_stat(self._h, name.encode(), C.byref(out)))
        return int(out.value)

    # ---- measurement ----------------------------------------------------------------------
    def profile_enable(self, on=True):
        self._check(self._L.snn_profile_enable(self._h, int(on)))

    def profile_reset(self):
        self._check(self._L.snn_profile_reset(self._h))

    def profile_read(self):
        n, ms = C.c_uint64(), C.c_double()
        self._check(self._L.snn_profile_read(self._h, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def profile_read_plasticity(self):
        """(steps measured, summed ms) of the plasticity launches (spike compaction + weight updates)"""
        n, ms = C.c_uint64(), C.c_double()
        self._check(self._L.snn_profile_read_plasticity(self._h, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def set_synthetic_drive(self, seed, fraction, voltage):
        """benchmark / load-test input: before every step a pseudo-random `fraction` of the neurons is set to `voltage`"""
        self._check(self._L.snn_set_synthetic_drive(self._h, int(seed), float(fraction), float(voltage)))

    def input_kernel_bytes(self):
        v = C.c_uint64()
        self._check(self._L.snn_input_kernel_bytes(self._h, C.byref(v)))
        return v.value


def probe_math(which, x, device=0):
    """Evaluate the stepper's device functions (0 exp, 1 pow3, 2 pow4) on the GPU."""
    L = _lib.load()
    a = np.ascontiguousarray(x, dtype=np.float32)
    out = _out(a.shape, a.dtype)
    _lib.check(L.snn_probe_math(device, which, a.ctypes.data_as(_lib.f32p), out.ctypes.data_as(_lib.f32p), a.size))
    return out


def probe_math_bits(which, first, count, stride=1, y=0.0, device=0):
    """The same over float BIT PATTERNS first + i * stride formed on the device; which = 3: powf(x, y)."""
    L = _lib.load()
    out = _out(count, np.float32)
    _lib.check(L.snn_probe_math_bits(device, which, first & 0xFFFFFFFF, stride, float(y), out.ctypes.data_as(_lib.f32p),
                                     count))
    return out


def probe_bandwidth(nbytes=8 << 30, repeats=5, device=0):
    """(read-only GB/s, copy GB/s) of the device with the stepper's access shape"""
    L = _lib.load()
    r, c = C.c_double(), C.c_double()
    _lib.check(L.snn_probe_bandwidth(device, nbytes, repeats, C.byref(r), C.byref(c)))
    return r.value, c.value
