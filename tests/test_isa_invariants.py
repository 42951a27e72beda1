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
