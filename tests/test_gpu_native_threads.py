"""The threading promise of include/snn_amd.h -- distinct handles may be driven from distinct threads -- held by a NATIVE host:
tests/cpp/threads_host_test.cpp starts four std::threads, one handle each (dense with histories, sparse with Rate cells, plastic,
a handle of a library that carries a generated model), 200 rounds of run / get / set per thread, three concurrent repetitions;
every thread's digest of what it read equals the digest of its workload run alone, and every thread finds its own message in
snn_last_error.  (Python threads enter the library one at a time under the GIL's release points; std::threads do not.)
SNN_TEST_TSAN=1 also builds the host program with -fsanitize=thread and reports what the sanitizer says about the HOST side."""
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(tmp_path, extra=()):
    exe = tmp_path / ("threads_host_test" + ("_tsan" if extra else ""))
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-Wall", *extra, "-o", str(exe), os.path.join(ROOT, "tests", "cpp", "threads_host_test.cpp"),
                    "-ldl", "-pthread"], check=True)
    return exe


def test_four_native_threads_one_handle_each(tmp_path, snn):
    from snn_amd import _lib, modelgen
    from snn_amd.examples_dsl import IZH_DSL
    generated = _lib.build_custom(modelgen.parse_description(IZH_DSL))
    exe = build(tmp_path)
    r = subprocess.run([str(exe), _lib.LIB_PATH, generated, "200"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["mismatches"] == 0 and res["threads"] == 4 and res["rounds"] == 200 and len(set(res["digests"])) == 4
    if os.environ.get("SNN_TEST_TSAN") == "1":
        tsan = build(tmp_path, ("-fsanitize=thread",))
        env = dict(os.environ, TSAN_OPTIONS="report_bugs=1 halt_on_error=0 exitcode=0 ignore_noninstrumented_modules=1")
        t = subprocess.run([str(tsan), _lib.LIB_PATH, generated, "40"], capture_output=True, text=True, timeout=900, env=env)
        print("tsan exit", t.returncode, "warnings", t.stderr.count("WARNING: ThreadSanitizer"))
        print(t.stderr[-3000:])
