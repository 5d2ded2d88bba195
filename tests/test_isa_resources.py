"""Register budget of the one-launch run (k_run_resident, 1024 threads = 128 registers per lane): no instantiation the host
launches may spill a vector register or execute a scratch access -- a spill inside its spin-wait step loop is paid in every
step of every run.  Compiles the kernel header alone for gfx950 with -save-temps (hipcc cross-compiles without a GPU) and reads
the amdhsa.kernels notes of the assembly (tests/isa_metadata.py)."""
import os
import re
import subprocess

import pytest

import isa_metadata

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "spiking-neural-networks_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# <MODEL, REGISTERS, CELLS, CHEM> as launch_run_resident (csrc/snn_network_step.hpp) picks them
# (the fifth parameter, STDP = weight updates inside the run, exists for electrical networks of neurons only; the sixth, LEND =
# idle wavefronts take the transmitter chains, for networks of at most 256 rows with both kinds of synapse)
VARIANTS = ([(m, False, c, True, False, False) for m in range(8) for c in (False, True)] +
            [(m, False, c, False, False, False) for m in range(8) for c in (False, True)] +
            [(m, True, c, False, False, False) for m in (0, 1, 3, 4) for c in (False, True)] + [(0, True, False, True, False, False)] +
            [(m, False, False, False, True, False) for m in range(8)] + [(0, True, False, False, True, False)] +
            [(m, False, False, True, False, True) for m in range(8)] + [(0, True, False, True, False, True)])


@pytest.fixture(scope="module")
def assembly(tmp_path_factory):
    d = tmp_path_factory.mktemp("isa")
    src = d / "resident_only.hip"
    inst = "\n".join(f"template __global__ void snn::k_run_resident<{m}, {str(r).lower()}, {str(c).lower()}, {str(h).lower()}, {str(p).lower()}, "
                     f"{str(e).lower()}>(const snn::ResidentRunArgs);" for m, r, c, h, p, e in VARIANTS)
    src.write_text(f'#include "{ROOT}/include/snn_amd.h"\n#include "snn_kernels_misc.hpp"\n#include "snn_kernels_resident.hpp"\n{inst}\n')
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wno-unused-result",
                    "-Wno-pass-failed", "-save-temps", f"-I{CSRC}", "-o", "lib.so", src.name], cwd=d, check=True, capture_output=True)
    return str(d / "resident_only-hip-amdgcn-amd-amdhsa-gfx950.s")


def kernel_bodies(asm_path):
    """{mangled name: text between its label and .end_amdhsa_kernel}"""
    text = open(asm_path, errors="replace").read()
    return {m.group(1): m.group(2) for m in re.finditer(r"^(_ZN3snn14k_run_resident\w+):.*?\n(.*?)\.end_amdhsa_kernel", text, re.S | re.M)}


def test_no_run_resident_variant_spills(assembly):
    table = {k: v for k, v in isa_metadata.parse(assembly).items() if k.startswith("snn::k_run_resident<")}
    assert len(table) == len(VARIANTS)
    bad = {k: v for k, v in table.items() if v["vgpr_spill"] or v["vgpr"] + v["agpr"] > 128 or v["max_threads"] != 1024}
    assert not bad, bad


def test_no_run_resident_variant_touches_scratch(assembly):
    bodies = kernel_bodies(assembly)
    assert len(bodies) == len(VARIANTS)
    for name, body in bodies.items():
        hits = [l.strip() for l in body.splitlines() if re.match(r"\s*(scratch_|buffer_(load|store).*offen)", l)]
        assert not hits, (name, hits[:4])
