"""Disassembly checks of hand-placed instructions hipcc cannot reason about (CPU-safe: hipcc -S cross-compiles for gfx950).

attention.hip issues the first MFMA of every S chain from inline asm (C != D, which the builtin cannot express).  hipcc models
neither the hazards of an asm statement nor its memory effects (guide 5.7), so the two properties the kernel relies on are pinned on
the generated ISA: (i) the asm MFMA is preceded, inside its own statement, by the `s_nop 1` that covers a VALU write of its C
operand; (ii) the first instruction after the statement that touches the asm MFMA's destination registers is the chained MFMA that
takes them WHOLE as SrcC (the one consumer an XDL result may feed with no software wait states) - never a VALU / memory
instruction, which would need 12 wait states hipcc does not insert for an asm producer."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _regs(tok):
    """'v[10:25]' -> set(10..25); 'v7' -> {7}; anything else -> empty."""
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


@pytest.mark.parametrize("src", ["attention.hip"])
def test_asm_mfma_feeds_only_the_chained_mfma(src, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path / "a.s"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-Wno-unused-result",
                           os.path.join(ROOT, "dove_amd", "csrc", src), "-o", str(out)], stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    n_asm = 0
    for i, l in enumerate(lines):
        if "v_mfma_f32_32x32x16_bf16" not in l or i < 2:
            continue
        # an asm MFMA sits between ;;#ASMSTART and ;;#ASMEND
        block_start = None
        for k in range(i - 1, max(i - 8, -1), -1):
            if "#ASMEND" in lines[k]:
                break
            if "#ASMSTART" in lines[k]:
                block_start = k
                break
        if block_start is None:
            continue
        n_asm += 1
        body = [x.strip() for x in lines[block_start + 1:i]]
        assert any(x.startswith("s_nop 1") for x in body), f"asm MFMA at line {i} lost its leading s_nop 1"
        ops = [t.strip() for t in l.split("v_mfma_f32_32x32x16_bf16")[1].split(",")]
        dst = _regs(ops[0])
        assert len(dst) == 16 and dst.isdisjoint(_regs(ops[3])), "asm MFMA: D must be 16 registers distinct from C"
        # first later instruction that touches D
        for k in range(i + 1, min(i + 400, len(lines))):
            t = lines[k].strip()
            if not t or t.startswith(";") or t.startswith("."):
                continue
            toks = re.findall(r"v\[\d+:\d+\]|v\d+", t)
            if not any(_regs(x) & dst for x in toks):
                continue
            assert t.startswith("v_mfma_f32_32x32x16_bf16"), f"line {k}: `{t}` touches the asm MFMA's result before the chained MFMA"
            o2 = [x.strip() for x in t.split("v_mfma_f32_32x32x16_bf16")[1].split(",")]
            assert _regs(o2[0]) == dst and _regs(o2[3].split()[0]) == dst, f"line {k}: the chained MFMA must take D whole as SrcC and vDst: `{t}`"
            break
        else:
            raise AssertionError(f"asm MFMA at line {i}: no consumer found")
    assert n_asm >= 4, f"expected the asm MFMAs of the S chains, found {n_asm}"
