"""The trap for stray host writes (test infrastructure; C half: tests/guard/snn_guard.c).

`install(snn_amd, log_dir)` changes three things in the calling process:

* **Every host buffer the device library is handed lives in the arena.**  The ctypes entry points of every loaded libsnn_amd*.so
  are wrapped: an argument that is a pointer into a numpy array (`arr.ctypes.data_as(...)`) is replaced by a pointer to a fresh
  arena buffer holding the same bytes, whose last byte is the last byte before an inaccessible page; after the call the bytes go
  back into the caller's array (when they changed) and the buffer is RETIRED: every second one becomes inaccessible at once and
  for good (address ranges are never reused), so a write or read that arrives after the call returned -- a staging thread of the
  runtime that is not done when the stream synchronisation says so -- faults at the instruction that does it; the others are
  filled with a pattern, stay writable for the next `canary_window` calls and are looked at again before they become
  inaccessible: a late writer that does not go through the CPU's page tables (a DMA engine) leaves no fault but a changed pattern.
* **The oracle's arrays live in the arena and are read-only while a call into the device library is under way.**
  `oracle_binding.Net.__init__` moves its arrays into one arena region per container (retired when the last view of it dies).
* **A SIGSEGV / SIGBUS handler** (snn_guard.c) reports address, region, tag (entry point, argument, call number, test and seed),
  access kind, thread id and name, pc and the backtrace of the faulting thread into `<log_dir>/guard-<pid>.log`, asks
  faulthandler for the Python stacks of all threads, lets the access complete and returns.

`Guard.check()` looks at the canaries that are due and returns the reports gathered since the last call (faults are counted by
the C side: `Guard.faults()`)."""
import collections
import ctypes as C
import faulthandler
import os
import signal
import subprocess
import threading
import weakref

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "guard", "snn_guard.c")
LIB = os.path.join(HERE, "guard", "_build", "libsnn_guard.so")

ST_LIVE, ST_READONLY, ST_QUARANTINE, ST_CANARY = 1, 2, 3, 4
CANARY_BYTE = 0xE5
_lib = None


def build():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        tmp = LIB + ".tmp.%d" % os.getpid()
        subprocess.run(["gcc", "-O1", "-g", "-fno-omit-frame-pointer", "-shared", "-fPIC", "-pthread", "-Wall", "-o", tmp, SRC], check=True)
        os.replace(tmp, LIB)
    return LIB


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.snn_guard_init.argtypes = [C.c_uint64, C.c_uint64, C.c_char_p, C.c_int]
        L.snn_guard_alloc.argtypes = [C.c_uint64, C.c_char_p]
        L.snn_guard_alloc.restype = C.c_void_p
        for fn in ("snn_guard_slack_intact",):
            getattr(L, fn).argtypes = [C.c_void_p]
        L.snn_guard_set_state.argtypes = [C.c_void_p, C.c_int]
        L.snn_guard_retag.argtypes = [C.c_void_p, C.c_char_p]
        for fn in ("snn_guard_fault_count", "snn_guard_region_count", "snn_guard_bytes_reserved", "snn_guard_host_table_count"):
            getattr(L, fn).restype = C.c_uint64
        L.snn_guard_note.argtypes = [C.c_char_p]
        _lib = L
    return _lib


class Guard:
    """the arena of this process (one per process: the handler and the reservation are global)"""
    _instance = None

    def __init__(self, log_dir, reserve_bytes=1 << 40, max_regions=1 << 22, canary_window=64, canary_every=2, big=64 << 20):
        assert Guard._instance is None, "one arena per process"
        os.makedirs(log_dir, exist_ok=True)
        self.log_path = os.path.join(log_dir, "guard-%d.log" % os.getpid())
        self.L = lib()
        # the Python stacks of all threads after every report: the handler raises SIGUSR2, which faulthandler dumps on
        self._pylog = open(self.log_path, "a")
        faulthandler.register(signal.SIGUSR2, file=self._pylog, all_threads=True, chain=False)
        rc = self.L.snn_guard_init(reserve_bytes, max_regions, self.log_path.encode(), int(signal.SIGUSR2))
        if rc != 0:
            raise OSError(rc, "snn_guard_init")
        self.canary_window, self.canary_every, self.big = canary_window, canary_every, big
        self.context = ""                  # the campaign writes "test seed" here: part of every tag
        self.calls = 0
        self.retired = 0
        self.reports = []
        self._canaries = collections.deque()          # (call number, address, bytes, tag)
        self._oracle = {}                  # address -> tag of the live oracle regions
        self._depth = 0
        self._lock = threading.RLock()
        Guard._instance = self

    # ---- buffers --------------------------------------------------------------------------
    def alloc(self, nbytes, tag):
        """(address, ctypes buffer over it); None when the arena is used up (the caller then passes the plain pointer)"""
        p = self.L.snn_guard_alloc(int(nbytes), tag.encode()[:87])
        if not p:
            return None
        return p, (C.c_char * int(nbytes)).from_address(p)

    def array(self, shape, dtype, tag):
        """a numpy array in the arena; its region is retired when the last view of it is gone"""
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        got = self.alloc(max(n, 1), tag)
        if got is None:
            return np.zeros(shape, dtype)
        p, buf = got
        weakref.finalize(buf, self._retire_quiet, p)
        return np.frombuffer(buf, dtype=np.uint8, count=n).view(dtype).reshape(shape)

    def _retire_quiet(self, p):
        self._oracle.pop(p, None)
        self.L.snn_guard_set_state(p, ST_QUARANTINE)

    def retire(self, p, nbytes, tag, view):
        """a buffer whose call has returned: inaccessible from now on, or a canary for the next calls"""
        self.retired += 1
        if self.canary_every and self.retired % self.canary_every == 0:
            view[:] = CANARY_BYTE
            self.L.snn_guard_set_state(p, ST_CANARY)
            self._canaries.append((self.calls, p, nbytes, tag))
        else:
            self.L.snn_guard_set_state(p, ST_QUARANTINE)

    def _look_at_canaries(self, everything=False):
        while self._canaries and (everything or self.calls - self._canaries[0][0] >= self.canary_window):
            born, p, nbytes, tag = self._canaries.popleft()
            now = np.frombuffer((C.c_char * nbytes).from_address(p), np.uint8)
            bad = np.flatnonzero(now != CANARY_BYTE)
            if bad.size:
                rec = {"late_write_without_fault": tag, "address": hex(p), "bytes": nbytes, "changed_bytes": int(bad.size),
                       "first_byte": int(bad[0]), "now": now[bad[0]:bad[0] + 16].tolist(), "calls_later": self.calls - born}
                self.reports.append(rec)
                self.L.snn_guard_note(("CANARY CHANGED " + repr(rec)).encode())
            self.L.snn_guard_set_state(p, ST_QUARANTINE)

    # ---- the oracle's arrays ---------------------------------------------------------------
    def adopt_oracle(self, net, label):
        """moves every array of an oracle container into ONE arena region"""
        items = [(k, a) for k, a in net.arr.items() if isinstance(a, np.ndarray) and a.size and a.flags["C_CONTIGUOUS"]]
        if not items:
            return
        total = sum((a.nbytes + 63) & ~63 for _, a in items)
        if total > self.big * 8:
            return
        got = self.alloc(total, label)
        if got is None:
            return
        p, buf = got
        weakref.finalize(buf, self._retire_quiet, p)
        off = 0
        for k, a in items:
            v = np.frombuffer(buf, dtype=np.uint8, count=a.nbytes, offset=off).view(a.dtype).reshape(a.shape)
            v[...] = a
            net.arr[k] = v
            off += (a.nbytes + 63) & ~63
        self._oracle[p] = label

    def _enter_call(self):
        with self._lock:
            self._depth += 1
            if self._depth == 1:
                for p in list(self._oracle):
                    self.L.snn_guard_set_state(p, ST_READONLY)

    def _leave_call(self):
        with self._lock:
            self._depth -= 1
            if self._depth == 0:
                for p in list(self._oracle):
                    self.L.snn_guard_set_state(p, ST_LIVE)

    # ---- the library's entry points ---------------------------------------------------------
    def wrap_function(self, name, fn):
        guard = self

        def guarded(*args):
            with guard._lock:
                guard.calls += 1
                call = guard.calls
            proxies, new_args = [], list(args)
            for i, a in enumerate(args):
                src = getattr(a, "_arr", None)
                if not isinstance(a, C._Pointer) or not isinstance(src, np.ndarray):
                    continue
                if not (0 < src.nbytes <= guard.big) or not src.flags["C_CONTIGUOUS"] or C.cast(a, C.c_void_p).value != src.ctypes.data:
                    continue
                tag = f"{name} arg{i} call{call} {guard.context}"
                got = guard.alloc(src.nbytes, tag)
                if got is None:
                    continue
                p, buf = got
                view = np.frombuffer(buf, np.uint8)
                view[:] = src.reshape(-1).view(np.uint8)
                new_args[i] = C.cast(p, type(a))
                proxies.append((src, view, p, tag))
            guard._enter_call()
            try:
                return fn(*new_args)
            finally:
                guard._leave_call()
                for src, view, p, tag in proxies:
                    flat = src.reshape(-1).view(np.uint8)
                    if not np.array_equal(view, flat):
                        if src.flags["WRITEABLE"]:
                            flat[:] = view
                        else:
                            guard.reports.append({"library_wrote_to_a_read_only_argument": tag})
                    if guard.L.snn_guard_slack_intact(p) != 1:
                        rec = {"bytes_before_the_buffer_changed": tag, "address": hex(p)}
                        guard.reports.append(rec)
                        guard.L.snn_guard_note(("UNDERRUN " + repr(rec)).encode())
                    guard.retire(p, view.size, tag, view)
                if guard._canaries:
                    guard._look_at_canaries()
        guarded.__name__ = name
        guarded.argtypes, guarded.restype, guarded.unguarded = getattr(fn, "argtypes", None), getattr(fn, "restype", None), fn
        return guarded

    def wrap_library(self, cdll, names):
        for name in names:
            fn = getattr(cdll, name)
            if not hasattr(fn, "unguarded"):
                setattr(cdll, name, self.wrap_function(name, fn))
        return cdll

    # ---- results ----------------------------------------------------------------------------
    def faults(self):
        return int(self.L.snn_guard_fault_count())

    def check(self, everything=False):
        """looks at the canaries that are due (all of them with everything=True) and hands out the reports gathered so far"""
        self._look_at_canaries(everything)
        out, self.reports = self.reports, []
        return out

    def stats(self):
        return {"calls": self.calls, "buffers_retired": self.retired, "library_host_tables": int(self.L.snn_guard_host_table_count()),
                "regions": int(self.L.snn_guard_region_count()),
                "arena_bytes": int(self.L.snn_guard_bytes_reserved()), "faults": self.faults(), "log": self.log_path}


def install(snn_amd, log_dir, internal_tables=True, **options):
    """arms the calling process (see the module's docstring); returns the Guard.  internal_tables: the library's own host tables
    come from the arena as well (snn_debug_set_host_allocator)"""
    import oracle_binding as ob
    options_internal_tables = internal_tables
    guard = Guard._instance or Guard(log_dir, **options)
    L = snn_amd._lib
    names = list(L.SIGNATURES)
    declare = L._declare
    if not hasattr(declare, "unguarded"):
        def guarded_declare(path):
            cdll = guard.wrap_library(declare(path), names)
            if options_internal_tables:
                fn = cdll.snn_debug_set_host_allocator
                L.check(getattr(fn, "unguarded", fn)(C.cast(guard.L.snn_guard_host_alloc, C.c_void_p), C.cast(guard.L.snn_guard_host_release, C.c_void_p)), cdll)
            return cdll
        guarded_declare.unguarded = declare
        L._declare = guarded_declare
    for cdll in [L._lib] + list(L._custom_libs.values()):
        if cdll is not None:
            guard.wrap_library(cdll, names)
    # the library's OWN host tables -- the temporaries its getters download into -- live in the arena too: a transfer that lands
    # after the call that owned the table returned faults (every library loaded from now on as well: guarded_declare below)
    hooks = (C.cast(guard.L.snn_guard_host_alloc, C.c_void_p), C.cast(guard.L.snn_guard_host_release, C.c_void_p))

    def hand_over(cdll):
        fn = getattr(cdll, "snn_debug_set_host_allocator")
        fn = getattr(fn, "unguarded", fn)
        L.check(fn(*hooks), cdll)
    if options_internal_tables:
        for cdll in [L._lib] + list(L._custom_libs.values()):
            if cdll is not None:
                hand_over(cdll)
    init = ob.Net.__init__
    if not hasattr(init, "unguarded"):
        counter = [0]

        def arena_init(self, *a, **k):
            init(self, *a, **k)
            counter[0] += 1
            guard.adopt_oracle(self, f"oracle Net {counter[0]} ({self.n_neurons}+{self.n_cells}) {guard.context}")
        arena_init.unguarded = init
        ob.Net.__init__ = arena_init
    return guard
