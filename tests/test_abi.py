"""The C-ABI shared library loads and exports every symbol include/snn_amd.h declares; the Python
binding declares exactly that set; failures are loud (no CPU fallback).  No compute calls here."""
import ctypes
import os
import re

import pytest

import snn_amd
from snn_amd import _lib

HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "snn_amd.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(snn_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    names = declared_functions()
    assert len(names) >= 40
    lib = _lib.load()
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} is declared in snn_amd.h but not exported by libsnn_amd.so"
    assert sorted(_lib.SIGNATURES) == names, "python binding and header disagree on the entry points"
    assert lib.snn_abi_version() == 2


def test_header_cites_the_reference_interface():
    text = open(HEADER).read()
    for needle in ("gpu_lattices/mod.rs:496-511", "gpu_lattices/mod.rs:1081-1100", "gpu_lattices/mod.rs:3183-3212",
                   "iterate_and_spike/mod.rs:3156-3189", "graph/mod.rs", "error/mod.rs:221-238"):
        assert needle in text


def test_error_codes_mirror_gpu_error_order():
    text = open(HEADER).read()
    order = ["PROGRAM_COMPILE", "KERNEL_COMPILE", "BUFFER_CREATE", "BUFFER_WRITE", "BUFFER_READ", "WAIT",
             "GET_DEVICE", "QUEUE"]
    for i, n in enumerate(order, start=1):
        assert re.search(rf"SNN_ERR_{n}\s*=\s*{i}\b", text)


def test_no_device_fails_loudly_not_silently():
    """Without a GPU the library must report GetDeviceFailure (7), never fall back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(snn_amd.SnnError) as e:
        snn_amd.DeviceNetwork()
    assert e.value.code == 7 and "device" in str(e.value).lower()


def test_null_and_bad_arguments_return_codes():
    L = _lib.load()
    assert L.snn_network_destroy(None) == 0
    assert L.snn_network_create(0, 99, 0, 0, 0, ctypes.byref(_lib.H())) == 11      # SNN_ERR_BAD_ARG
    assert L.snn_network_create(0, 0, 0, 0, 0, None) == 11
    assert L.snn_run(None, 1) == 11
    assert b"null" in L.snn_last_error()


def test_product_never_touches_the_oracle():
    """Nothing under the product package may import / load / link anything from oracle/ or tests/."""
    pkg = os.path.dirname(_lib.__file__)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                for needle in ("oracle/", "snn_oracle", "oracle_binding", "libsnn_oracle", "snn_o_"):
                    assert needle not in src, f"{f} references the oracle ({needle})"


def test_comm_lifecycle_is_refused_while_the_collectives_are_replaced():
    """snn_set_collectives swaps the seven per-step entry points only: a communicator then is the host's own object, and the
    three RCCL life-cycle calls must fail with SNN_ERR_BAD_STATE instead of handing it to ncclCommDestroy (or calling through a
    null pointer when librccl is absent).  No device call is made."""
    L = _lib.load()
    q = _lib.COMM_QUERY_FN(lambda comm, out: 0)
    table = _lib.Collectives(q, q, _lib.ALL_GATHER_FN(lambda *a: 0), _lib.SEND_RECV_FN(lambda *a: 0),
                             _lib.SEND_RECV_FN(lambda *a: 0), _lib.GROUP_FN(lambda: 0), _lib.GROUP_FN(lambda: 0))
    assert L.snn_set_collectives(ctypes.byref(table)) == 0
    try:
        ident = ctypes.create_string_buffer(128)
        fake = ctypes.c_int(0)
        out = ctypes.c_void_p()
        assert L.snn_comm_unique_id(ident) == 12   # SNN_ERR_BAD_STATE
        assert L.snn_comm_init_rank(ident, 1, 0, 0, ctypes.byref(out)) == 12
        assert L.snn_comm_destroy(ctypes.cast(ctypes.pointer(fake), ctypes.c_void_p)) == 12
        world, rank = ctypes.c_int(-1), ctypes.c_int(-1)
        assert L.snn_comm_count(ctypes.cast(ctypes.pointer(fake), ctypes.c_void_p), ctypes.byref(world), ctypes.byref(rank)) == 0
    finally:
        assert L.snn_set_collectives(None) == 0


# ---- the three statements of the boundary agree: the header, the ctypes binding, the Rust stub of INTEGRATION.md -------------
INTEGRATION = os.path.join(os.path.dirname(HEADER), "..", "INTEGRATION.md")

_C_SCALARS = {"int": ctypes.c_int, "uint32_t": ctypes.c_uint32, "uint64_t": ctypes.c_uint64, "size_t": ctypes.c_size_t,
              "float": ctypes.c_float, "double": ctypes.c_double, "int32_t": ctypes.c_int32, "uint8_t": ctypes.c_uint8,
              "int64_t": ctypes.c_int64}
_RUST_TO_C = {"c_int": "int", "u32": "uint32_t", "u64": "uint64_t", "usize": "size_t", "f32": "float", "f64": "double", "i32": "int32_t",
              "u8": "uint8_t", "c_char": "char", "c_void": "void", "SnnNetwork": "snn_network_t", "SnnExchangePlan": "snn_exchange_plan",
              "SnnCollectives": "snn_collectives"}


def header_prototypes():
    """name -> (return type, [parameter types]) of every function the header declares; types as C text without names / spaces"""
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    text = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", "", text, flags=re.S)        # (function-pointer members are not prototypes)
    out = {}
    for ret, name, params in re.findall(r"([A-Za-z_][\w\s\*]*?)\b(snn_[a-z0-9_]+)\s*\(([^()]*)\)\s*;", text):
        types = []
        for p in [q.strip() for q in params.split(",")]:
            if p in ("void", ""):
                continue
            m = re.match(r"(.*?)(\b[A-Za-z_]\w*)?$", p)          # the trailing identifier is the parameter's name
            t = m.group(1) if (m.group(2) and (m.group(1).strip())) else p
            types.append(t.replace(" ", ""))
        out[name] = (ret.replace("extern", "").replace(" ", ""), types)
    return out


def rust_prototypes():
    """the `fn snn_*` declarations inside the extern "C" blocks of INTEGRATION.md, translated to the header's spelling"""
    text = open(INTEGRATION).read()
    out = {}
    for block in re.findall(r"extern \"C\" \{(.*?)\n\s*\}", text, flags=re.S):
        block = re.sub(r"//[^\n]*", "", block)
        for name, params, ret in re.findall(r"fn\s+(snn_[a-z0-9_]+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+))?;", block, flags=re.S):
            def c_type(t):
                t = t.strip()
                stars = ""
                const = False
                while t.startswith("*"):
                    kind, t = t.split(None, 1)
                    const = const or (kind == "*const" and stars == "")      # constness of the pointee of the innermost pointer is what C spells
                    inner_const = kind == "*const"
                    stars += "*"
                    t = t.strip()
                    last_const = inner_const
                base = _RUST_TO_C.get(t, t)
                if stars:
                    return ("const" if last_const else "") + base + stars
                return base
            types = [c_type(p.split(":", 1)[1]) for p in params.split(",") if ":" in p]
            out[name] = (c_type(ret) if ret else "void", types)
    return out


def test_rust_stub_of_the_integration_document_matches_the_header():
    """INTEGRATION.md's extern "C" declarations are what a maintainer of the reference would paste: name, arity and every
    scalar / pointer type must be the header's (the document cannot be compiled here -- no rustc -- so it is parsed)"""
    head, rust = header_prototypes(), rust_prototypes()
    assert len(rust) >= 25, f"only {len(rust)} declarations found in INTEGRATION.md"
    for name, (ret, types) in rust.items():
        assert name in head, f"INTEGRATION.md declares {name}, which the header does not"
        hret, htypes = head[name]
        assert ret == hret, f"{name}: returns {ret} in INTEGRATION.md, {hret} in the header"
        assert len(types) == len(htypes), f"{name}: {len(types)} parameters in INTEGRATION.md, {len(htypes)} in the header"
        for i, (a, b) in enumerate(zip(types, htypes)):
            # a `*mut T` may stand where C says `const T *` only for the opaque handle (Rust has no const handle type in the stub)
            if a != b and not (a.replace("const", "") == b.replace("const", "") and "snn_network_t" in a):
                raise AssertionError(f"{name}, parameter {i}: {a} in INTEGRATION.md, {b} in the header")


def test_ctypes_signatures_match_the_header():
    head = header_prototypes()
    for name, (res, args) in _lib.SIGNATURES.items():
        hret, htypes = head[name]
        assert len(args) == len(htypes), f"{name}: {len(args)} ctypes arguments, {len(htypes)} header parameters"
        want_ret = _C_SCALARS.get(hret)
        if want_ret is not None:
            assert res is want_ret, f"{name}: returns {hret}, bound as {res}"
        else:
            assert res in (ctypes.c_char_p, ctypes.c_void_p), f"{name}: returns {hret}, bound as {res}"
        for i, (a, t) in enumerate(zip(args, htypes)):
            if "*" in t or "snn_exchange_fn" in t or "snn_host_alloc_fn" in t or "snn_host_release_fn" in t:
                assert a in (ctypes.c_void_p, ctypes.c_char_p) or hasattr(a, "_type_") or hasattr(a, "contents"), \
                    f"{name}, parameter {i}: {t} bound as {a}"
            else:
                assert a is _C_SCALARS[t], f"{name}, parameter {i}: {t} bound as {a}"


def test_every_option_and_statistic_is_documented_in_the_header():
    """snn_set_option / snn_get_stat names are part of the boundary: each name the library accepts appears, quoted, in the header"""
    src = open(os.path.join(os.path.dirname(_lib.__file__), "csrc", "snn_network.hip")).read()
    head = open(HEADER).read()
    quoted = set(re.findall(r'"([a-z][a-z_0-9]+)"', head))
    a, b = src.index("int snn_set_option"), src.index("int snn_get_stat")
    options = set(re.findall(r'n == "([a-z_0-9]+)"', src[a:b]))
    stats = set(re.findall(r'n == "([a-z_0-9]+)"', src[b:b + 12000]))
    assert len(options) >= 25 and len(stats) >= 20
    # (the header spells the run_timing_* family once: "run_timing_poll" / "_barrier" / "_turns" / "_update")
    stats -= {"run_timing_barrier", "run_timing_turns", "run_timing_update"}
    assert not (options - quoted), f"options the header does not name: {sorted(options - quoted)}"
    assert not (stats - quoted), f"statistics the header does not name: {sorted(stats - quoted)}"


def test_every_exported_definition_is_a_function_try_block():
    """The exception barrier of the C ABI (csrc/snn_network_state.hpp, abi_exception): every entry point the header declares is
    defined in snn_network.hip as `... snn_x(...) ABI_TRY { ... } ABI_CATCH` -- a function-try-block whose handler turns any C++
    exception into a status code.  Checked on the text of the translation unit: no definition may be added without it."""
    src = open(os.path.join(os.path.dirname(_lib.__file__), "csrc", "snn_network.hip")).read()
    state = open(os.path.join(os.path.dirname(_lib.__file__), "csrc", "snn_network_state.hpp")).read()
    assert "#define ABI_TRY try" in state and "catch (...) { return abi_exception(__func__); }" in state
    for name in declared_functions():
        m = re.search(rf"^(?:int|const char \*)\s*{name}\(", src, flags=re.M)
        assert m, f"{name} is declared in the header but not defined at the top level of snn_network.hip"
        depth, j = 0, m.end() - 1
        while True:                                   # the end of the parameter list
            depth += {"(": 1, ")": -1}.get(src[j], 0)
            if depth == 0:
                break
            j += 1
        assert src[j + 1:].lstrip().startswith("ABI_TRY"), f"{name}: the definition does not open with ABI_TRY"
        # the body's closing brace (braces in string literals do not occur unbalanced in this file) is followed by the handler
        k = src.index("{", j)
        depth = 0
        while True:
            depth += {"{": 1, "}": -1}.get(src[k], 0)
            if depth == 0:
                break
            k += 1
        want = "ABI_CATCH_PTR" if src[m.start():m.end()].startswith("const char") else "ABI_CATCH"
        tail = src[k + 1:k + 40].lstrip()
        assert tail.startswith(want) and (want == "ABI_CATCH_PTR" or not tail.startswith("ABI_CATCH_PTR")), \
            f"{name}: the definition does not end with {want}"
    # and nothing else in the library may use std::vector with the default allocator (every host table counts as an allocation of the
    # failure hook, snn_debug_fail_alloc_at)
    for f in ("snn_network.hip", "snn_network_state.hpp", "snn_network_step.hpp", "snn_network_exchange.hpp"):
        text = re.sub(r"//.*", "", open(os.path.join(os.path.dirname(_lib.__file__), "csrc", f)).read())
        assert text.count("std::vector<") == (1 if f == "snn_network_state.hpp" else 0), f
