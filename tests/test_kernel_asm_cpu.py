"""Static checks on the device code of libamlike.so (no GPU needed).

nnet_pnorm_rows_kernel and nnet_out_rows_kernel issue their matrix instructions and their LDS
fragment reads as inline assembly, which the compiler does not understand: it knows neither that a matrix instruction
delivers its result tens of cycles after issue and fetches its operands late, nor that a register
filled by such an LDS read is only valid after the wave's counted wait.  Any register-to-register
copy the compiler inserts around them (it did, in an earlier form of the output-layer kernel: the
accumulators were copied right behind the matrix instruction that wrote them) silently reads stale
data.  The kernels are written so that no such copy is needed; this test holds the built code to it."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "kaldi_amd", "csrc", "libamlike.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def disassemble_kernel(name_part):
    if not (os.path.exists(LIB) and os.path.exists(OBJDUMP)):
        pytest.skip("library or llvm-objdump missing")
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        lib = shutil.copy(LIB, os.path.join(tmp, "lib.so"))
        subprocess.run([OBJDUMP, "--offloading", lib], cwd=tmp, capture_output=True, check=True)
        objs = [f for f in os.listdir(tmp) if "gfx950" in f]
        assert objs, "no gfx950 code object in the library"
        text = ""
        for o in objs:
            text += subprocess.run([OBJDUMP, "-d", os.path.join(tmp, o)], capture_output=True, text=True,
                                   check=True).stdout
    m = re.search(r"^[0-9a-f]+ <(\S*%s\S*)>:\n(.*?)(?=^[0-9a-f]+ <|\Z)" % re.escape(name_part), text, re.S | re.M)
    assert m, "kernel %s not found in the library" % name_part
    return [l.split("//")[0].strip() for l in m.group(2).splitlines() if l.strip()]


def regs(operand):
    """The vector registers an operand names: 'v[18:33]' -> {18..33}, 'v45' -> {45}, else empty."""
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", operand)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", operand)
    return {int(m.group(1))} if m else set()


@pytest.mark.parametrize("kernel,steps,parked", [("nnet_pnorm_rows_kernelILi13ELi8E", 3, 8),
                                                 ("nnet_pnorm_rows_kernelILi23ELi4E", 5, 8),
                                                 ("nnet_out_rows_kernel", 7, 32)])
def test_nothing_touches_a_matrix_instructions_registers_right_behind_it(kernel, steps, parked):
    """For every matrix instruction and the six instructions behind it: no vector instruction reads
    or writes its result registers, and none writes its operands' registers (an LDS read may name
    them as its destination: its data arrives long after).  `kernel`: a part of the mangled name
    (the template arguments included where several instantiations exist); `steps`: the kernel is
    known to hold at least 39 x steps matrix instructions; `parked`: moves to and from the accumulation registers
    tolerated up to the last matrix instruction (values parked across a loop)."""
    code = disassemble_kernel(kernel)
    assert sum(l.startswith("v_mfma") for l in code) >= 39 * steps
    # (a handful of index values parked in the accumulation registers across the loop are harmless;
    # moves of matrix results or fragments would number dozens and are caught below anyway)
    last_mfma = max(i for i, l in enumerate(code) if l.startswith("v_mfma"))
    assert sum(l.startswith("v_accvgpr") for l in code[:last_mfma]) <= parked  # (an epilogue may use the dead fragments' registers)
    assert not any(l.startswith("scratch_") for l in code)  # no spills
    bad = []
    for i, l in enumerate(code):
        if not l.startswith("v_mfma"):
            continue
        ops = [o.strip() for o in l.split(None, 1)[1].split(",")]
        dst, src = regs(ops[0]), regs(ops[1]) | regs(ops[2])
        for later in code[i + 1:i + 7]:
            if later.startswith("v_mfma") or not later.startswith("v_"):
                continue
            lops = [o.strip() for o in later.split(None, 1)[1].split(",")]
            written, read = regs(lops[0]), set().union(*[regs(o) for o in lops[1:]]) if len(lops) > 1 else set()
            if (written | read) & dst or written & src:
                bad.append((l, later))
    assert not bad, "vector instructions too close behind a matrix instruction: %s" % bad[:3]
