"""ctypes binding of the C ABI declared in include/snn_amd.h.

The shared library is built in-tree (csrc/libsnn_amd.so) by `build()` below or by
`__graft_entry__.build()`.  There is NO CPU fallback: if the library is missing or
does not load, importing callers get a loud `SnnLibraryError`.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("SNN_AMD_LIB", os.path.join(CSRC, "libsnn_amd.so"))   # override: A/B builds
HEADER = os.path.join(os.path.dirname(_HERE), "include", "snn_amd.h")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
               "-Wall", "-Wno-unused-result", "-Wno-pass-failed"]


class SnnLibraryError(RuntimeError):
    pass


class SnnError(RuntimeError):
    """A failing C-ABI call; `.code` is the snn_status (1..8 mirror the reference's GPUError)."""

    def __init__(self, code, message):
        super().__init__(f"snn_amd error {code}: {message}")
        self.code = code


def _hipcc():
    return os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


_hipcc_version_cache = {}


def hipcc_version():
    """first line of `hipcc --version` that names the compiler (cached)"""
    exe = _hipcc()
    if exe not in _hipcc_version_cache:
        try:
            out = subprocess.run([exe, "--version"], capture_output=True, text=True, timeout=60).stdout
            lines = [ln.strip() for ln in out.splitlines() if "version" in ln.lower()]
            _hipcc_version_cache[exe] = "; ".join(lines[:2]) or "unknown"
        except Exception:                       # noqa: BLE001
            _hipcc_version_cache[exe] = "unknown"
    return _hipcc_version_cache[exe]


def source_hash(extra_files=(), defines=()):
    """sha256 over what a library is compiled from: every .hip / .hpp of csrc/ (sorted by name), the header, the generated
    header of a custom library, the flags and the compiler's version -- the CONTENT decides whether a built library is
    current, not a time stamp (an edit made while hipcc runs would otherwise pass as built)"""
    import hashlib
    h = hashlib.sha256()
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".hpp"))] + [HEADER] + list(extra_files)
    for path in files:
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    h.update(" ".join(HIPCC_FLAGS + list(defines)).encode())
    h.update(hipcc_version().encode())
    return h.hexdigest()


def _build_one(out, cmd_tail, digest, force, what):
    """compiles `out` unless a library built from exactly these sources is already there; one line of provenance on stderr"""
    import sys
    stamp = out + ".hash"
    have = open(stamp).read().strip() if os.path.exists(stamp) else None
    if not force and os.path.exists(out) and have == digest:
        print(f"[snn_amd build] reused {what}: {os.path.relpath(out)} source_sha256={digest[:16]} hipcc=\"{hipcc_version()}\"", file=sys.stderr)
        return out
    why = "forced" if force else ("no library" if not os.path.exists(out) else ("no hash on record" if have is None else "sources changed"))
    if not force and os.path.exists(out) and have is not None and hipcc_version() == "unknown":
        # no compiler on this box (a pushed, prebuilt library): the version is part of the hash, so it can never match -- the
        # library on disk is all there is; say so instead of failing in a compile that cannot start
        print(f"[snn_amd build] reused {what} WITHOUT a compiler to check it against: {os.path.relpath(out)} recorded={have[:16]} "
              f"(hipcc not found: {_hipcc()})", file=sys.stderr)
        return out
    tmp = out + ".tmp.%d" % os.getpid()        # (concurrent builders -- campaign workers -- never write each other's file)
    subprocess.run([_hipcc()] + HIPCC_FLAGS + cmd_tail + ["-o", tmp, os.path.join(CSRC, "snn_network.hip")], check=True, cwd=CSRC)
    # the sources may have been edited while the compiler ran: the hash on record is the one taken BEFORE the compile
    # started only if they still hash the same now; otherwise the library is kept but marked stale (no hash)
    os.replace(tmp, out)
    after = source_hash(*_hash_args.get(out, ((), ())))
    if after == digest:
        with open(stamp, "w") as f:
            f.write(digest + "\n")
    elif os.path.exists(stamp):
        os.remove(stamp)
    print(f"[snn_amd build] compiled {what} ({why}): {os.path.relpath(out)} source_sha256={digest[:16]} hipcc=\"{hipcc_version()}\""
          + ("" if after == digest else " -- SOURCES CHANGED DURING THE COMPILE: not recorded as current"), file=sys.stderr)
    return out


_hash_args = {}


def build(force=False):
    """Compile every HIP source for gfx950 into csrc/libsnn_amd.so (hipcc cross-compiles without a GPU).  Reuses the library
    only when the hash of its sources (source_hash) is the one on record next to it (libsnn_amd.so.hash)."""
    _hash_args[LIB_PATH] = ((), ())
    return _build_one(LIB_PATH, [], source_hash(), force, "libsnn_amd.so")


def build_custom(model, force=False):
    """Compile a library that carries the generated neuron model `model` (modelgen.NeuronModel) as SNN_MODEL_CUSTOM --
    or every block of a modelgen.Description: neuron, spike train (SNN_ST_CUSTOM), refractoriness (kind 2):
    csrc/generated/<name>.hpp + csrc/generated/libsnn_amd_<name>.so.  Returns the library path."""
    from . import modelgen
    import hashlib
    gen = os.path.join(CSRC, "generated")
    os.makedirs(gen, exist_ok=True)
    text = modelgen.hip_source(model)
    # the file name carries the content: two descriptions that share a type name never share (or overwrite) a library
    stem = f"{model.name}_{hashlib.sha1(text.encode()).hexdigest()[:10]}"
    header = os.path.join(gen, stem + ".hpp")
    out = os.path.join(gen, f"libsnn_amd_{stem}.so")
    if force or not os.path.exists(header) or open(header).read() != text:
        with open(header, "w") as f:
            f.write(text)
    define = '-DSNN_CUSTOM_MODEL_HEADER="generated/%s.hpp"' % stem
    _hash_args[out] = ((header,), (define,))
    return _build_one(out, [define], source_hash((header,), (define,)), force, f"custom library {stem}")


class ExchangePlan(C.Structure):
    """snn_exchange_plan of include/snn_amd.h"""
    _fields_ = [("mode", C.c_int32), ("n_shards", C.c_uint32), ("shard_index", C.c_uint32), ("shard_stride", C.c_uint32),
                ("planes", C.c_uint32), ("plane_id", C.c_uint32 * 4), ("send", C.c_void_p), ("recv", C.c_void_p),
                ("send_words", C.c_uint64), ("recv_words", C.c_uint64)]


EXCHANGE_ALLGATHER, EXCHANGE_HALO = 0, 1

EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)      # snn_exchange_fn(user, hip_stream)

# snn_collectives of include/snn_amd.h (snn_set_collectives)
COMM_QUERY_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int))
ALL_GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p)
SEND_RECV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p)
GROUP_FN = C.CFUNCTYPE(C.c_int)


class Collectives(C.Structure):
    _fields_ = [("comm_count", COMM_QUERY_FN), ("comm_user_rank", COMM_QUERY_FN), ("all_gather", ALL_GATHER_FN),
                ("send", SEND_RECV_FN), ("recv", SEND_RECV_FN), ("group_start", GROUP_FN), ("group_end", GROUP_FN)]


f32p = C.POINTER(C.c_float)
u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)
u8p = C.POINTER(C.c_uint8)
u64p = C.POINTER(C.c_uint64)
H = C.c_void_p   # snn_network_t*

# name -> (restype, argtypes); kept in one place so tests can check it against the header
SIGNATURES = {
    "snn_abi_version": (C.c_int, []),
    "snn_last_error": (C.c_char_p, []),
    "snn_custom_model": (C.c_char_p, []),
    "snn_custom_spike_train": (C.c_char_p, []),
    "snn_custom_refractoriness": (C.c_char_p, []),
    "snn_custom_neurotransmitter_kinetics": (C.c_char_p, []),
    "snn_custom_receptor_kinetics": (C.c_char_p, []),
    "snn_custom_receptors": (C.c_char_p, []),
    "snn_network_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(H)]),
    "snn_network_destroy": (C.c_int, [H]),
    "snn_network_add_lattice": (C.c_int, [H, C.c_uint32, C.c_uint32, C.c_uint32]),
    "snn_network_add_spike_train_lattice": (C.c_int, [H, C.c_uint32, C.c_uint32, C.c_uint32]),
    "snn_network_finalize": (C.c_int, [H]),
    "snn_network_finalize_shard": (C.c_int, [H, C.c_uint32, C.c_uint32]),
    "snn_network_finalize_shard_by_lattice": (C.c_int, [H, C.c_uint32, C.c_uint32]),
    "snn_shard_ranges": (C.c_int, [H, u32p, u32p, C.c_uint32, u32p]),
    "snn_network_sizes": (C.c_int, [H, u32p, u32p, u32p, u32p]),
    "snn_network_lattice_range": (C.c_int, [H, C.c_uint32, u32p, u32p]),
    "snn_set_attr_f32": (C.c_int, [H, C.c_uint32, C.c_char_p, f32p, C.c_size_t]),
    "snn_get_attr_f32": (C.c_int, [H, C.c_uint32, C.c_char_p, f32p, C.c_size_t]),
    "snn_set_attr_u32": (C.c_int, [H, C.c_uint32, C.c_char_p, u32p, C.c_size_t]),
    "snn_get_attr_u32": (C.c_int, [H, C.c_uint32, C.c_char_p, u32p, C.c_size_t]),
    "snn_set_attr_i32": (C.c_int, [H, C.c_uint32, C.c_char_p, i32p, C.c_size_t]),
    "snn_get_attr_i32": (C.c_int, [H, C.c_uint32, C.c_char_p, i32p, C.c_size_t]),
    "snn_set_graph_dense": (C.c_int, [H, f32p, u32p, C.c_size_t]),
    "snn_get_graph_dense": (C.c_int, [H, f32p, u32p, C.c_size_t]),
    "snn_set_graph_rows": (C.c_int, [H, C.c_uint32, C.c_uint32, f32p, u32p]),
    "snn_get_graph_rows": (C.c_int, [H, C.c_uint32, C.c_uint32, f32p, u32p]),
    "snn_fill_graph_synthetic": (C.c_int, [H, C.c_uint64, C.c_float, C.c_float, C.c_int]),
    "snn_network_use_csr": (C.c_int, [H, C.c_int]),
    "snn_set_graph_csr": (C.c_int, [H, u64p, u32p, f32p, C.c_uint64]),
    "snn_get_graph_csr": (C.c_int, [H, f32p, C.c_uint64]),
    "snn_set_synapses": (C.c_int, [H, C.c_int, C.c_int]),
    "snn_set_plasticity": (C.c_int, [H, C.c_uint32] + [C.c_float] * 5 + [C.c_int]),
    "snn_set_history": (C.c_int, [H, C.c_int, C.c_int]),
    "snn_reset_history": (C.c_int, [H]),
    "snn_get_clock": (C.c_int, [H, u64p]),
    "snn_set_clock": (C.c_int, [H, C.c_uint64]),
    "snn_set_spike_train_clock": (C.c_int, [H, C.c_uint32, C.c_uint64]),
    "snn_get_spike_train_clock": (C.c_int, [H, C.c_uint32, u64p]),
    "snn_reset_timing": (C.c_int, [H]),
    "snn_run": (C.c_int, [H, C.c_uint64]),
    "snn_step_begin": (C.c_int, [H]),
    "snn_step_end": (C.c_int, [H]),
    "snn_step_begin_local": (C.c_int, [H]),
    "snn_refresh_begin": (C.c_int, [H, C.POINTER(C.c_int)]),
    "snn_refresh_end": (C.c_int, [H]),
    "snn_exchange_plan_get": (C.c_int, [H, C.c_void_p]),
    "snn_exchange_peers": (C.c_int, [H, u64p, u64p, u64p, u64p]),
    "snn_halo_needs": (C.c_int, [H, C.c_uint32, u32p, C.c_uint32, u32p]),
    "snn_cells_read": (C.c_int, [H, u32p, C.c_uint32, u32p]),
    "snn_halo_set_sends": (C.c_int, [H, C.c_uint32, u32p, C.c_uint32]),
    "snn_halo_commit": (C.c_int, [H]),
    "snn_comm_unique_id": (C.c_int, [C.c_void_p]),
    "snn_comm_init_rank": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "snn_comm_destroy": (C.c_int, [C.c_void_p]),
    "snn_comm_count": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "snn_set_connection_kind": (C.c_int, [H, C.c_uint32, C.c_uint32, C.c_int]),
    "snn_set_pending_rows": (C.c_int, [H, C.c_uint32, C.c_uint32, f32p]),
    "snn_get_pending_rows": (C.c_int, [H, C.c_uint32, C.c_uint32, f32p]),
    "snn_set_counter_rows": (C.c_int, [H, C.c_uint32, C.c_uint32, u8p]),
    "snn_get_counter_rows": (C.c_int, [H, C.c_uint32, C.c_uint32, u8p]),
    "snn_p2p_local": (C.c_int, [H, u64p, u64p, u64p, u64p, u64p]),
    "snn_p2p_connect": (C.c_int, [H, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64]),
    "snn_p2p_commit": (C.c_int, [H]),
    "snn_p2p_ipc_export": (C.c_int, [H, C.c_void_p]),
    "snn_p2p_ipc_import": (C.c_int, [C.c_int, C.c_void_p, u64p, u64p, u64p]),
    "snn_p2p_ipc_close": (C.c_int, [C.c_int, C.c_uint64, C.c_uint64, C.c_uint64]),
    "snn_set_collectives": (C.c_int, [C.c_void_p]),
    "snn_comm_exchange_halo_lists": (C.c_int, [H, C.c_void_p]),
    "snn_exchange": (C.c_int, [H, C.c_void_p]),
    "snn_run_sharded": (C.c_int, [H, C.c_void_p, C.c_uint64]),
    "snn_run_sharded_custom": (C.c_int, [H, C.c_void_p, C.c_void_p, C.c_uint64]),
    "snn_exchange_noop": (C.c_int, [C.c_void_p, C.c_void_p]),
    "snn_stream": (C.c_int, [H, C.POINTER(C.c_void_p)]),
    "snn_set_stream": (C.c_int, [H, C.c_void_p]),
    "snn_synchronize": (C.c_int, [H]),
    "snn_history_steps": (C.c_int, [H, u64p]),
    "snn_get_voltage_history": (C.c_int, [H, C.c_uint32, f32p, C.c_size_t]),
    "snn_get_spike_history": (C.c_int, [H, C.c_uint32, u8p, C.c_size_t]),
    "snn_set_bcm": (C.c_int, [H, C.c_uint32, C.c_float, C.c_float, C.c_float, C.c_int]),
    "snn_set_reward_modulator": (C.c_int, [H, C.c_uint32] + [C.c_float] * 8 + [C.c_int]),
    "snn_get_dopamine": (C.c_int, [H, C.c_uint32, f32p]),
    "snn_apply_reward": (C.c_int, [H, C.c_float]),
    "snn_run_with_reward": (C.c_int, [H, C.c_float]),
    "snn_set_trace_rows": (C.c_int, [H, C.c_uint32, C.c_uint32, f32p]),
    "snn_get_trace_rows": (C.c_int, [H, C.c_uint32, C.c_uint32, f32p]),
    "snn_set_traces_csr": (C.c_int, [H, f32p, C.c_uint64]),
    "snn_get_traces_csr": (C.c_int, [H, f32p, C.c_uint64]),
    "snn_set_pending_csr": (C.c_int, [H, f32p, C.c_uint64]),
    "snn_get_pending_csr": (C.c_int, [H, f32p, C.c_uint64]),
    "snn_set_counters_csr": (C.c_int, [H, u8p, C.c_uint64]),
    "snn_get_counters_csr": (C.c_int, [H, u8p, C.c_uint64]),
    "snn_set_firing_times": (C.c_int, [H, C.c_uint32, u32p, f32p, C.c_size_t]),
    "snn_set_graph_history": (C.c_int, [H, C.c_uint32, C.c_int]),
    "snn_get_graph_history": (C.c_int, [H, C.c_uint32, f32p, C.c_size_t]),
    "snn_set_history_stride": (C.c_int, [H, C.c_uint32]),
    "snn_set_reduced_history": (C.c_int, [H, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float]),
    "snn_get_average_voltage_history": (C.c_int, [H, C.c_uint32, f32p, C.c_size_t]),
    "snn_get_eeg_history": (C.c_int, [H, C.c_uint32, f32p, C.c_size_t]),
    "snn_get_spike_counts": (C.c_int, [H, C.c_uint32, u32p, C.c_size_t]),
    "snn_set_option": (C.c_int, [H, C.c_char_p, C.c_int]),
    "snn_get_stat": (C.c_int, [H, C.c_char_p, u64p]),
    "snn_debug_verify_report": (C.c_char_p, [H]),
    "snn_debug_checkpoint": (C.c_int, [H, C.c_int]),
    "snn_debug_fail_alloc_at": (C.c_int, [C.c_int64, u64p]),
    "snn_debug_set_host_allocator": (C.c_int, [C.c_void_p, C.c_void_p]),
    "snn_profile_enable": (C.c_int, [H, C.c_int]),
    "snn_profile_reset": (C.c_int, [H]),
    "snn_profile_read": (C.c_int, [H, u64p, C.POINTER(C.c_double)]),
    "snn_profile_read_plasticity": (C.c_int, [H, u64p, C.POINTER(C.c_double)]),
    "snn_set_synthetic_drive": (C.c_int, [H, C.c_uint64, C.c_float, C.c_float]),
    "snn_input_kernel_bytes": (C.c_int, [H, u64p]),
    "snn_probe_bandwidth": (C.c_int, [C.c_int, C.c_uint64, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "snn_probe_math": (C.c_int, [C.c_int, C.c_int, f32p, f32p, C.c_size_t]),
    "snn_probe_math_bits": (C.c_int, [C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_float, f32p, C.c_size_t]),
}

_lib = None


_custom_libs = {}


def load(path=None):
    """Load csrc/libsnn_amd.so (or, with `path`, a library built by build_custom) and declare every entry point;
    raises SnnLibraryError if absent."""
    global _lib
    if path is not None:
        path = os.path.abspath(path)
        if path not in _custom_libs:
            load()                                   # torch / HIP runtime ordering as for the default library
            _custom_libs[path] = _declare(path)
        return _custom_libs[path]
    if _lib is not None:
        return _lib
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7; if it is loaded AFTER
    # the system copy this library links to, the process ends up with two runtimes and torch sees no
    # GPU.  Importing torch first makes both resolve to the same (torch's) runtime, which is also what
    # RCCL inside torch.distributed must share with the exchange buffer (parallel.py).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    _lib = _declare(LIB_PATH)
    return _lib


def _declare(path):
    if not os.path.exists(path):
        raise SnnLibraryError(
            f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(the stepper has no CPU fallback)")
    try:
        lib = C.CDLL(path)
    except OSError as e:
        raise SnnLibraryError(f"cannot load {path}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise SnnLibraryError(f"{path} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    return lib


def check(code, lib=None):
    if code != 0:
        msg = (lib or load()).snn_last_error()
        raise SnnError(code, msg.decode("utf-8", "replace") if msg else "")
