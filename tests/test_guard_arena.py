"""The trap for stray host writes (tests/guard_arena.py, tests/guard/snn_guard.c) catches what it is there for, on the CPU, with a
misbehaving "library" (the helpers at the end of snn_guard.c): a store that arrives after the call returned is reported with the
thread and the function that made it; a getter that overruns its buffer faults at the first byte past the end; a write into the
oracle's arrays during a library call is reported as such; a late write that no fault can see changes a canary; and a well-behaved
call reports nothing.  Runs in a child process: the handler and the address-space reservation are global to a process."""
import json
import os
import subprocess
import sys
import textwrap

HERE = os.path.dirname(os.path.abspath(__file__))

CHILD = textwrap.dedent("""
    import ctypes as C, gc, json, os, sys, time
    sys.path.insert(0, sys.argv[1])
    import numpy as np
    import guard_arena as ga
    import oracle_binding as ob
    import parity

    out_dir = sys.argv[2]
    guard = ga.Guard(out_dir, canary_window=4)
    L = guard.L
    f32p, u32p = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
    L.snn_guard_test_getter.argtypes = [f32p, C.c_uint64, C.c_float, C.c_uint64, C.c_int, C.POINTER(C.c_ulong)]
    L.snn_guard_test_join.argtypes = [C.c_ulong]
    L.snn_guard_test_overrun.argtypes = [f32p, C.c_uint64, C.c_uint64]
    L.snn_guard_test_stray.argtypes = [C.c_void_p, C.c_uint32]
    L.snn_guard_test_setter.argtypes = [f32p, C.c_uint64]
    L.snn_guard_test_setter.restype = C.c_float
    guard.wrap_library(L, ["snn_guard_test_getter", "snn_guard_test_overrun", "snn_guard_test_stray", "snn_guard_test_setter"])
    res = {}

    # the oracle's arrays move into the arena
    init = ob.Net.__init__
    def arena_init(self, *a, **k):
        init(self, *a, **k)
        guard.adopt_oracle(self, "oracle Net under test")
    ob.Net.__init__ = arena_init
    net = parity.make_oracle(parity.Layout([(0, 4, 5)], [(3, 2, 2)]), st_kind=ob.ST_POISSON)
    v0 = net["current_voltage"].copy()
    net.run(5)                                           # the oracle itself works on arena memory
    res["oracle_ran"] = bool((net["current_voltage"] != v0).any())

    # 1. a well-behaved setter and getter: values arrive, nothing is reported
    guard.context = "case well-behaved"
    x = np.arange(100, dtype=np.float32)
    res["setter_sum"] = float(L.snn_guard_test_setter(x.ctypes.data_as(f32p), x.size))
    net["current_voltage"] = np.float32(-60.0)           # the test's own write between two library calls
    res["setter_sum_of_oracle_array"] = float(L.snn_guard_test_setter(net["current_voltage"].ctypes.data_as(f32p), net.n_neurons))
    res["faults_after_good_calls"] = guard.faults()

    # 2. a store that arrives 150 ms after the getter returned, into a buffer that was retired to "inaccessible"
    guard.context = "case late-store"
    guard.retired = 0                                    # (next retirement: number 1, not a canary)
    out = np.zeros(16, np.float32)
    t = C.c_ulong()
    L.snn_guard_test_getter(out.ctypes.data_as(f32p), out.size, 2.5, 7, 150, C.byref(t))
    res["getter_values"] = out.tolist()
    L.snn_guard_test_join(t)
    res["faults_after_late_store"] = guard.faults()
    res["getter_values_after_the_late_store"] = out.tolist()

    # 3. the same into a buffer that was retired to "canary": no fault, the pattern changed
    guard.context = "case late-store-canary"
    out2 = np.zeros(16, np.float32)
    L.snn_guard_test_getter(out2.ctypes.data_as(f32p), out2.size, 2.5, 7, 100, C.byref(t))
    L.snn_guard_test_join(t)
    res["faults_after_canary_store"] = guard.faults()
    res["canary_reports"] = guard.check(everything=True)

    # 4. a getter that writes 3 floats past its buffer
    guard.context = "case overrun"
    out3 = np.zeros(8, np.float32)
    L.snn_guard_test_overrun(out3.ctypes.data_as(f32p), out3.size, 3)
    res["faults_after_overrun"] = guard.faults()
    res["overrun_values"] = out3.tolist()

    # 5. the library scribbles over an oracle array while a call is under way
    guard.context = "case stray-write"
    L.snn_guard_test_stray(net["nt_flags"].ctypes.data + 38 * 4, 0xFFFFFFFE)
    res["faults_after_stray"] = guard.faults()
    res["nt_flags_word"] = int(net["nt_flags"].reshape(-1)[38])
    net["nt_flags"].reshape(-1)[38] = 0                  # ... and the test may write it again afterwards
    res["faults_after_own_write"] = guard.faults()

    # 6. a dead oracle container's memory becomes inaccessible; a view keeps it alive
    keep = net["v_th"]
    addr = keep.ctypes.data
    del net
    gc.collect()
    res["view_still_readable"] = float(keep[0])
    res["faults_with_view_alive"] = guard.faults()
    del keep
    gc.collect()
    L.snn_guard_test_stray(addr, 5)                      # a write to the dead container
    res["faults_after_write_to_dead_oracle"] = guard.faults()
    res["stats"] = guard.stats()
    print(json.dumps(res))
""")


def test_the_trap_names_the_writer(tmp_path):
    script = tmp_path / "child.py"
    script.write_text(CHILD)
    p = subprocess.run([sys.executable, str(script), HERE, str(tmp_path / "log")], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    res = json.loads(p.stdout.strip().splitlines()[-1])
    assert res["oracle_ran"]
    assert res["setter_sum"] == 4950.0 and res["setter_sum_of_oracle_array"] == -60.0 * 20
    assert res["faults_after_good_calls"] == 0
    # the late store: reported once, the caller's array is untouched by it (it went to the retired buffer)
    assert res["getter_values"] == [2.5] * 16 and res["getter_values_after_the_late_store"] == [2.5] * 16
    assert res["faults_after_late_store"] == 1
    assert res["faults_after_canary_store"] == 1 and len(res["canary_reports"]) == 1
    c = res["canary_reports"][0]
    assert "snn_guard_test_getter arg0" in c["late_write_without_fault"] and "case late-store-canary" in c["late_write_without_fault"]
    assert c["first_byte"] == 7 * 4 and c["changed_bytes"] >= 1
    assert res["faults_after_overrun"] == 2 and res["overrun_values"] == [3.0] * 8
    assert res["faults_after_stray"] == 3 and res["nt_flags_word"] == 0xFFFFFFFE and res["faults_after_own_write"] == 3
    assert res["view_still_readable"] == 30.0 and res["faults_with_view_alive"] == 3
    assert res["faults_after_write_to_dead_oracle"] == 4
    log = open(res["stats"]["log"]).read()
    faults = log.split("=== snn_guard fault ")[1:]
    assert len(faults) == 4
    late, overrun, stray, dead = faults
    assert "access WRITE" in late and "state retired" in late and "snn_guard_test_getter arg0" in late and "case late-store" in late
    assert "offset 28" in late and "name late-writer" in late and "snn_guard_test_late_store" in late     # who: thread and function
    assert "GUARD PAGE, 0 bytes past the end" in overrun and "case overrun" in overrun and "snn_guard_test_overrun" in overrun
    assert "state read-only" in stray and "oracle Net under test" in stray and "snn_guard_test_stray" in stray
    assert "state retired" in dead and "oracle Net under test" in dead
    assert "Current thread" in log or "Thread 0x" in log or "Stack (most recent call first)" in log     # faulthandler's Python stacks
