"""The BVH box step's node fetch leaves loads IN FLIGHT across compiler-scheduled code (rayz_device.hpp, bvh_node_step: the
inline asm waits for the left child's loads only, a second asm waits for the right child's): the compiler must not read,
copy or overwrite the registers those loads are still writing.  The source holds them with "+v" operands; this test reads
the gfx950 ISA of every kernel that contains the step and checks that nothing between the two waits names them."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _kernels(asm: str):
    out, name, body = {}, None, []
    for line in asm.split("\n"):
        m = re.match(r"^(_ZN8rayz_dev\w*trace_kernel_bvh\w*):", line)
        if m:
            name, body = m.group(1), []
            continue
        if name is not None:
            if line.startswith(".Lfunc_end"):
                out[name] = body
                name = None
            elif not line.lstrip().startswith(";"):
                body.append(line)
    return out


def test_no_instruction_touches_the_node_fetch_registers_in_flight(tmp_path):
    from rayz_amd import _build

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    asm = tmp_path / "dev.s"
    flags = [f for f in _build.HIPFLAGS if f not in ("-fPIC", "-Wall", "-Wextra")]
    subprocess.run([hipcc, *flags, "--cuda-device-only", "-S", "-o", str(asm), os.path.join(ROOT, "rayz_amd", "csrc", "rayz_hip.hip")],
                   check=True, capture_output=True, timeout=600)
    kernels = _kernels(asm.read_text())
    assert len(kernels) >= 4, sorted(kernels)  # f32 / f64 x two record formats
    blocks = 0
    for name, L in kernels.items():
        for i, line in enumerate(L):
            m = re.search(r"s_waitcnt vmcnt\((\d)\) lgkmcnt\((\d)\)", line)
            if not m or m.group(1) != m.group(2) or m.group(1) not in "12":
                continue
            n = int(m.group(1))  # loads still in flight after this wait: the last n of the fetch (global and LDS turn alike)
            dests, j = [], i - 1
            while j > 0 and i - j < 40:
                mm = re.search(r"(?:global_load_dwordx4|ds_read_b128) v\[(\d+):(\d+)\]", L[j])
                if mm:
                    dests.append((int(mm.group(1)), int(mm.group(2))))
                j -= 1
            assert len(dests) == 4 * n, (name, i, dests)  # n pairs... both turns issue 2n loads each
            regs = set()
            for a, b in dests[:n] + dests[2 * n:3 * n]:  # the last n of either turn
                regs |= set(range(a, b + 1))
            k = i + 1
            while "s_waitcnt vmcnt(0) lgkmcnt(0)" not in L[k]:
                for a, b, c in re.findall(r"v\[(\d+):(\d+)\]|\bv(\d+)\b", L[k]):
                    used = {int(c)} if c else set(range(int(a), int(b) + 1))
                    assert not (used & regs), f"{name}: `{L[k].strip()}` touches a register of a node load still in flight"
                k += 1
                assert k - i < 80, (name, "no second wait after the split one")
            blocks += 1
    assert blocks >= 2 * 4  # two steps per wave-level decision in each one-path kernel


def test_product_kernels_stay_off_their_codegen_cliffs(tmp_path):
    """Properties of the compiled gfx950 code that a refactor can lose without any test noticing a wrong pixel (only a slower frame):
    no vector register of a trace kernel is spilled to scratch in the f32 kernels (the BVH kernels sit at 123 – 128 VGPRs, 4 waves per
    SIMD), the BVH kernels' shading pass fetches its tables with global, not flat, loads (round 4: behind the empty asm that pins the
    table pointers the compiler no longer knows the address space unless it is told), the retired experiment kernels are not in the
    product build, and the flat-list kernel's scalar spills stay where round 4 left them (its scan loop lives on an allocation edge)."""
    from rayz_amd import _build

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    asm = tmp_path / "dev.s"
    flags = [f for f in _build.HIPFLAGS if f not in ("-fPIC", "-Wall", "-Wextra")]
    subprocess.run([hipcc, *flags, "--cuda-device-only", "-S", "-o", str(asm), os.path.join(ROOT, "rayz_amd", "csrc", "rayz_hip.hip")],
                   check=True, capture_output=True, timeout=600)
    text = asm.read_text()
    assert "trace_kernel_bvh2" not in text and "trace_kernel_bvhx" not in text
    meta = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n((?:\s+\.\w+:.*\n)+)", text):
        fields = dict(re.findall(r"\.(\w+):\s+(\S+)", m.group(2)))
        meta[m.group(1)] = fields
    traces = {k: v for k, v in meta.items() if "trace_kernel" in k}
    assert len(traces) == 6, sorted(traces)  # flat f32 / f64, BVH f32 / f64 x two node-record formats
    for name, f in traces.items():
        f64 = "IdL" in name  # (TraceArgs<double>)
        assert int(f["vgpr_count"]) <= 128, (name, f["vgpr_count"])
        assert int(f["vgpr_spill_count"]) <= (1 if f64 else 0), (name, f["vgpr_spill_count"])
        if "bvh" not in name and not f64:
            assert int(f["sgpr_spill_count"]) <= 80, (name, f["sgpr_spill_count"])  # 70 in round 4 (63 in round 3)
    bodies = _kernels(text)
    for name, L in bodies.items():
        assert not any(re.match(r"\s*flat_load", l) for l in L), f"{name}: a flat_load is back in the BVH kernel (shade's table pointers lost their address space)"
        assert not any(re.match(r"\s*scratch_(load|store)", l) for l in L) or "IdL" in name, f"{name}: scratch traffic in an f32 BVH kernel"


def test_experiments_build_still_compiles(tmp_path):
    """The retired experiment kernels (two paths per lane, round 3; walker / shader waves, round 4) live behind -DRAYZ_EXPERIMENTS with
    their tools (tools/build_experiments.sh): they are evidence, and evidence that no longer compiles cannot be re-measured.  Device code
    only, all three shader-wave counts of the exchange kernel are instantiable."""
    from rayz_amd import _build

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    flags = [f for f in _build.HIPFLAGS if f not in ("-fPIC", "-Wall", "-Wextra")]
    asm = tmp_path / "x.s"
    subprocess.run([hipcc, *flags, "-DRAYZ_EXPERIMENTS", "-DRAYZ_BVHX_SHADERS=5", "--cuda-device-only", "-S", "-o", str(asm),
                    os.path.join(ROOT, "rayz_amd", "csrc", "rayz_hip.hip")], check=True, capture_output=True, timeout=600)
    text = asm.read_text()
    assert "trace_kernel_bvh2" in text and "trace_kernel_bvhx" in text
