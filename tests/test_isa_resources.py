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


# ---- the closing input pass (k_inputs_dense_close): the neuron update shares the streaming pass's register allocation -----------
CLOSE_VARIANTS = [(m, chem, shape, nt) for m in range(8) for (chem, nt) in ((False, 3), (True, 1), (True, 3)) for shape in (1, 2)]


@pytest.fixture(scope="module")
def close_assembly(tmp_path_factory):
    d = tmp_path_factory.mktemp("isa_close")
    src = d / "close_only.hip"
    inst = "\n".join(f"template __global__ void snn::k_inputs_dense_close<{m}, true, {str(c).lower()}, {s}, {nt}>(const snn::DenseStepArgs);"
                     for m, c, s, nt in CLOSE_VARIANTS)
    plain = "\n".join(f"template __global__ void snn::k_inputs_dense<true, {str(c).lower()}, {s}, {nt}, 0>(const snn::InputsArgs);"
                      for c, nt in ((False, 3), (True, 1), (True, 3)) for s in (1, 2))
    src.write_text(f'#include "{ROOT}/include/snn_amd.h"\n#include "snn_kernels_dense_step.hpp"\n{inst}\n{plain}\n')
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-c", "--cuda-device-only", "-Wno-unused-result",
                    "-Wno-pass-failed", "-save-temps", f"-I{CSRC}", "-o", "close.o", src.name], cwd=d, check=True, capture_output=True)
    return str(d / "close_only-hip-amdgcn-amd-amdhsa-gfx950.s")


def test_the_closing_pass_keeps_the_occupancy_of_the_plain_pass(close_assembly):
    """wavefronts per SIMD = 512 // registers: the closing form of a pass must not run fewer than the plain pass does (three for the
    4-column shape, four for the 2-column shape), and nothing may spill"""
    table = isa_metadata.parse(close_assembly)
    close = {k: v for k, v in table.items() if k.startswith("snn::k_inputs_dense_close<")}
    assert len(close) == len(CLOSE_VARIANTS)
    for name, r in close.items():
        assert r["vgpr_spill"] == 0 and r["scratch"] == 0, (name, r)
        m, _, chem, shape, nt = [x.strip() for x in name[name.index("<") + 1:name.rindex(">")].split(",")]
        plain = table[f"snn::k_inputs_dense<true, {chem}, {shape}, {nt}, 0>"]
        want = min(512 // plain["vgpr"], 3 if shape == "1" else 4)
        assert 512 // (r["vgpr"] + r["agpr"]) >= want, (name, r["vgpr"], "plain", plain["vgpr"])


# ---- the sparse step over the step image (k_step_csr_img): full occupancy, windows by LDS-DMA, header by scalar loads ---------------
@pytest.fixture(scope="module")
def image_assembly(tmp_path_factory):
    d = tmp_path_factory.mktemp("isa_image")
    src = d / "image_only.hip"
    inst = "\n".join(f"template __global__ void snn::k_step_csr_img<{m}, false>(const snn::CsrStepArgs);" for m in range(8))
    src.write_text(f'#include "{ROOT}/include/snn_amd.h"\n#include "snn_kernels_csr.hpp"\n{inst}\n'
                   "template __global__ void snn::k_step_csr<0, true, false, false>(const snn::CsrStepArgs);\n")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-c", "--cuda-device-only", "-Wno-unused-result",
                    "-Wno-pass-failed", "-save-temps", f"-I{CSRC}", "-o", "image.o", src.name], cwd=d, check=True, capture_output=True)
    return str(d / "image_only-hip-amdgcn-amd-amdhsa-gfx950.s")


def test_the_image_step_keeps_eight_wavefronts_per_simd_and_stages_by_lds_dma(image_assembly):
    table = isa_metadata.parse(image_assembly)
    img = {k: v for k, v in table.items() if k.startswith("snn::k_step_csr_img<")}
    assert len(img) == 8
    for name, r in img.items():
        # 64 registers = 8 wavefronts per SIMD, 16 KiB of LDS per workgroup of 4 wavefronts = 128 KiB per CU at that occupancy
        # (Hodgkin-Huxley's update, its exponentials' table loads in flight together, needs 98 with or without the image: five
        # wavefronts per SIMD, as its plain step)
        limit = 102 if name == "snn::k_step_csr_img<2, false>" else 64
        assert r["vgpr_spill"] == 0 and r["scratch"] == 0 and r["vgpr"] + r["agpr"] <= limit and r["lds"] == 16384, (name, r)
    text = open(image_assembly, errors="replace").read()
    body = re.search(r"^_ZN3snn14k_step_csr_imgILi0ELb0EEEvNS_11CsrStepArgsE:.*?\n(.*?)\.end_amdhsa_kernel", text, re.S | re.M).group(1)
    assert len(re.findall(r"global_load_lds_dword\b", body)) == 16           # one LDS-DMA load per window piece, no ds_write pass
    assert len(re.findall(r"s_load_dwordx16", body)) >= 2                     # the slice header's pieces arrive in scalar registers
    assert len(re.findall(r"global_load_dwordx4", body)) >= 8                 # records of two entries: 16 bytes per lane and load
    plain = table["snn::k_step_csr<0, true, false, false>"]
    assert plain["lds"] == 0 and plain["vgpr"] <= 64                          # the plain step is untouched


# ---- the one-launch step over four wavefronts per chunk (k_step_resident_q) and the argument warm-up (warm_kernel_arguments) --------
@pytest.fixture(scope="module")
def quarter_assembly(tmp_path_factory):
    d = tmp_path_factory.mktemp("isa_quarters")
    src = d / "quarters_only.hip"
    inst = "\n".join(f"template __global__ void snn::k_step_resident_q<{m}, true, {c}>(const snn::ResidentArgs);" for m in range(8) for c in ("true", "false"))
    src.write_text(f'#include "{ROOT}/include/snn_amd.h"\n#include "snn_kernels_misc.hpp"\n#include "snn_kernels_resident.hpp"\n{inst}\n'
                   "template __global__ void snn::k_update<2, true>(const snn::UpdateArgs);\n")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-c", "--cuda-device-only", "-Wno-unused-result",
                    "-Wno-pass-failed", "-save-temps", f"-I{CSRC}", "-o", "quarters.o", src.name], cwd=d, check=True, capture_output=True)
    return str(d / "quarters_only-hip-amdgcn-amd-amdhsa-gfx950.s")


def test_the_quarter_step_fits_two_chunks_and_its_arguments_arrive_in_one_round_trip(quarter_assembly):
    table = isa_metadata.parse(quarter_assembly)
    q = {k: v for k, v in table.items() if k.startswith("snn::k_step_resident_q<")}
    assert len(q) == 16
    for name, r in q.items():
        # 512 threads = 8 wavefronts on a CU, two per SIMD: 256 registers each; 64 weights + 64 products per lane are half of that
        assert r["vgpr_spill"] == 0 and r["scratch"] == 0 and r["vgpr"] + r["agpr"] <= 256 and r["max_threads"] == 512, (name, r)
    text = open(quarter_assembly, errors="replace").read()
    for label, nbytes in (("_ZN3snn17k_step_resident_qILi0ELb1ELb1EEEvNS_12ResidentArgsE", 1640), ("_ZN3snn8k_updateILi2ELb1EEEvNS_10UpdateArgsE", 1368)):
        body = re.search(rf"^{label}:.*?\n(.*?)\.end_amdhsa_kernel", text, re.S | re.M).group(1)
        # ONE inline-assembly statement: a word of every 64-byte line of the arguments, then the wait -- nothing of the compiler's
        # between the loads and the wait (the scratch registers are written when a load returns)
        block = re.search(r";;#ASMSTART\n(.*?);;#ASMEND", body, re.S).group(1)
        lines = [l.strip() for l in block.strip().splitlines()]
        loads = [l for l in lines if l.startswith("s_load_dword ")]
        assert lines[-1] == "s_waitcnt lgkmcnt(0)" and len(loads) == len(lines) - 1, lines[-3:]
        offsets = sorted(int(l.rsplit(",", 1)[1], 0) for l in loads)
        want = (nbytes + 63) // 64
        assert offsets == [64 * i for i in range(want)], (label, offsets[-3:], want)
        assert offsets[-1] + 4 <= nbytes                     # never past the explicit arguments: a kernel without implicit ones has
                                                             # nothing behind them (.kernarg_segment_size = sizeof the struct)
        notes = text[:re.search(rf"\.name:\s*{label}\s", text).start()]
        size = int(re.findall(r"\.kernarg_segment_size:\s*(\d+)", notes)[-1])       # (the entry's fields are in alphabetical order)
        assert offsets[-1] + 4 <= size, (label, size)
        # and it is the first thing the kernel does with memory: no vector load in front of it
        assert not re.search(r"global_load|buffer_load", body[:body.index(";;#ASMSTART")])
