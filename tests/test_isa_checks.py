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


def _kernel_body(lines, mangled_prefix):
    """Lines of the first kernel whose label starts with `mangled_prefix`, up to its s_endpgm."""
    start = next(i for i, l in enumerate(lines) if l.startswith(mangled_prefix + ":"))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    return lines[start:end + 1]


@pytest.fixture(scope="module")
def igemm_asm(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "g.s"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-Wno-unused-result",
                           os.path.join(ROOT, "dove_amd", "csrc", "igemm.hip"), "-o", str(out)], stderr=subprocess.DEVNULL)
    return open(out).read()


def _bare_vmcnt_waits(walk):
    in_asm, bare = False, []
    for l in walk:
        if "#ASMSTART" in l:
            in_asm = True
        elif "#ASMEND" in l:
            in_asm = False
        elif "s_waitcnt" in l and "vmcnt" in l and not in_asm:
            bare.append(l.strip())
    return bare


@pytest.mark.parametrize("variant,taps,per_step", [("ILb0ELb0ELb1ELb0ELb1ELi0ELb0E", 9, 64), ("ILb1ELb0ELb1ELb0ELb1ELi0ELb0E", 9, 64), ("ILb0ELb0ELb1ELb1ELb1ELi0ELb0E", 4, 64),
                                                    ("ILb0ELb0ELb1ELb0ELb1ELi1ELb0E", 9, 64)],
                         ids=["direct", "upsample-in-addressing", "sub-pixel", "tiles-32x16"])
def test_halo4x_one_wave_per_simd_budget_and_counted_waits(igemm_asm, variant, taps, per_step):
    """conv3x3_halo4x runs ONE wave per SIMD on the whole 512-register file: its 8 x 8 accumulator tile of 16 x 16 blocks is the 256 AGPRs
    (asm MFMAs with the accumulator tied and constrained to the AGPR file - as builtins the allocator rotated the quads through VGPRs and
    spilled), everything else must fit the 256 VGPRs WITHOUT scratch and without a single accumulator move inside the tap walk.  Its operand
    ring is kept in flight by hand-counted `s_waitcnt vmcnt(n)` in asm; a wait the compiler adds on its own between the first and the last
    MFMA means it lost track of the queue and drains the ring there (what happened to gemm8p in round 3).  The unrolled body is two groups
    of `taps` spatial-tap steps x 64 MFMAs of v_mfma_f32_16x16x32_bf16 (9 taps; 4 in the sub-pixel form of the upsample-fused conv,
    dove_conv_desc.w_sub), each step with its 16 fragment reads pinned between MFMA pairs.  The tile-geometry variant (kPart 1: the last
    tile column of an image that ends within its first half, walked in 32 x 16 tiles) is the same walk with another block -> pixel map and
    LDS image: the same budget and counts."""
    text = igemm_asm
    name = f"_Z21conv3x3_halo4x_kernel{variant}Ev9IgemmArgs"

    def prop(key):
        m = re.search(rf"\.set {name}\.{key}, (\d+)", text)
        assert m, f"{name}: no {key} record"
        return int(m.group(1))

    assert prop("private_seg_size") == 0, f"{name}: scratch in use"
    assert prop("num_agpr") == 4 * per_step and prop("num_vgpr") <= 256, f"{name}: register budget {prop('num_vgpr')} + {prop('num_agpr')}"
    body = _kernel_body(text.split("\n"), name)
    assert not any("scratch_" in l for l in body), f"{name}: scratch instructions"
    mf = [i for i, l in enumerate(body) if "v_mfma_f32_16x16x32_bf16" in l]
    assert len(mf) == 2 * taps * per_step, f"{name}: expected 2 groups x {taps} taps x {per_step} MFMAs, found {len(mf)}"
    assert not any("v_mfma_f32_32x32x16_bf16" in l for l in body), f"{name}: a 32x32x16 MFMA in the 16x16x32 walk"
    walk = body[mf[0]:mf[-1]]
    bare = _bare_vmcnt_waits(walk)
    assert not bare, f"{name}: compiler-inserted vmcnt waits inside the tap walk: {bare}"
    moves = [l.strip() for l in walk if "v_accvgpr" in l]
    assert not moves, f"{name}: accumulator moves inside the tap walk: {moves[:4]}"
    # every MFMA accumulates in place in the AGPR file (vDst == SrcC)
    for i in mf:
        ops_ = [t.strip() for t in body[i].split("v_mfma_f32_16x16x32_bf16")[1].split(",")]
        assert ops_[0].startswith("a[") and ops_[0] == ops_[3].split()[0], f"{name}: `{body[i].strip()}`"
    reads = 16 if per_step == 64 else 12
    assert sum("ds_read_b128" in l for l in walk) >= 2 * taps * reads - 12, f"{name}: {reads} fragment reads per step expected"


def test_gemm8p_k_walk_keeps_its_operand_stream_in_flight(igemm_asm):
    """gemm8p stages its operands one K-64 step ahead by LDS-DMA and waits for them with hand-placed `s_waitcnt vmcnt(0)` at the two
    places the ring argument needs (igemm.hip: end of LOAD / MFMA of the odd phases).  hipcc's own wait insertion knows nothing about
    those asm waits: when the epilogue's VGPR loads looked pending to it at the K loop's header, it put `s_waitcnt vmcnt(1)` /
    `vmcnt(0)` in front of the first fragment reads of every 128-deep chunk and drained the stream there (round 3; fixed by ending the
    epilogue with the `s_waitcnt` BUILTIN, which its tracking sees).  Pinned here: inside the K walk every vmcnt wait sits in an asm
    block, each phase has its 12 fragment reads and 32 MFMAs of v_mfma_f32_16x16x32_bf16 (two waves per SIMD: 256 registers each, the
    128 accumulator registers are VGPRs), and the kernel uses no scratch."""
    text = igemm_asm
    lines = text.split("\n")
    for variant in ("ILb0ELb0ELb0ELb1E", "ILb1ELb0ELb0ELb1E", "ILb0ELb1ELb0ELb1E"):          # plain, GELU, gated
        name = f"_Z13gemm8p_kernel{variant}Ev9IgemmArgsx"
        m = re.search(rf"\.set {name}\.private_seg_size, (\d+)", text)
        assert m and int(m.group(1)) == 0, f"{name}: scratch in use"
        m = re.search(rf"\.set {name}\.num_vgpr, (\d+)", text)
        assert m and int(m.group(1)) <= 256, f"{name}: {m.group(1)} VGPRs: two waves per SIMD no longer fit"
        body = _kernel_body(lines, name)
        # the K walk = the stretch between the first and the last MFMA of the kernel (the epilogue has none)
        mf = [i for i, l in enumerate(body) if "v_mfma_f32_16x16x32_bf16" in l]
        assert len(mf) == 128, f"{name}: expected 4 phases x 32 MFMAs in the unrolled chunk, found {len(mf)}"
        assert not any("v_mfma_f32_32x32x16_bf16" in l for l in body), f"{name}: a 32x32x16 MFMA in the 16x16x32 phases"
        first_read = next(i for i, l in enumerate(body) if "ds_read_b128" in l)      # the prologue only stages; the epilogue's reads follow the last MFMA
        walk = body[first_read - 2:mf[-1] + 12]
        bare = _bare_vmcnt_waits(walk)
        assert not bare, f"{name}: compiler-inserted vmcnt waits inside the K walk would drain the operand stream: {bare}"
        assert sum("ds_read_b128" in l for l in walk) == 48, f"{name}: 12 fragment reads per phase expected"
        assert sum("lds" in l and "buffer_load_dwordx4" in l for l in walk) == 16, f"{name}: 8 LDS-DMAs in each of the two even phases expected"
        assert not any("v_accvgpr" in l for l in walk), f"{name}: accumulator moves inside the K walk"


def test_no_product_kernel_spills(igemm_asm, tmp_path):
    """Every kernel of the product library (the translation units csrc/build.sh compiles for libdove_hip.so) fits its registers: no scratch
    segment, no scratch instructions.  A spill would not fail any parity test - only the clock."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    build = open(os.path.join(ROOT, "dove_amd", "csrc", "build.sh")).read()
    srcs = re.search(r'^SRCS="([^"]+)"', build, re.M).group(1).split()
    assert "igemm" in srcs and "attention" in srcs, srcs
    procs = []
    for f in srcs:
        if f == "igemm":
            continue                                            # compiled once by the fixture
        out = tmp_path / f"{f}.s"
        procs.append((f, out, subprocess.Popen([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-Wno-unused-result",
                                                os.path.join(ROOT, "dove_amd", "csrc", f + ".hip"), "-o", str(out)], stderr=subprocess.DEVNULL)))
    texts = {"igemm": igemm_asm}
    for f, out, pr in procs:
        assert pr.wait() == 0, f"{f}.hip did not compile"
        texts[f] = open(out).read()
    n = 0
    for f, text in texts.items():
        for name, size in re.findall(r"\.set (\S+)\.private_seg_size, (\d+)", text):
            n += 1
            assert int(size) == 0, f"{f}.hip: {name} uses {size} bytes of scratch"
        assert "scratch_load" not in text and "scratch_store" not in text, f"{f}.hip: scratch instructions"
    assert n >= 60, f"only {n} kernel records found - did the assembly format change?"


def test_pipelined_attention_step_is_what_was_placed(tmp_path):
    """attention_pipe.hip places every instruction of its K walk by hand (asm MFMAs / exps / adds / packs / LDS reads, sched_barrier fences) and
    owns hazards hipcc does not model for asm producers (guide 5.7).  Pinned on the ISA of a MAIN-LOOP step (between two workgroup barriers):
    the instruction multiset that was placed (per query block and step 16 MFMAs, 32 exps, 32 single adds, 16 packs; 16 fragment reads into AGPRs, 4 LDS-DMAs, no
    accumulator moves, no packed-f32 VALU - an anti-lever beside MFMAs that -O3 SLP-packing produces from plain adds), and the distance the
    pipeline promises between an S chain's last MFMA and the first VALU instruction that reads its registers (>= 60 instructions, >= 30 in
    the one-block form: the 18 wait states a 16-pass XDL result needs are covered with margin)."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path / "ap.s"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-Wno-unused-result",
                           os.path.join(ROOT, "dove_amd", "csrc", "attention_pipe.hip"), "-o", str(out)], stderr=subprocess.DEVNULL)
    lines = [l.strip() for l in open(out).read().split("\n")]
    all_lines = lines
    # NB = 2: the kernel as described (two query blocks per wave); NB = 1: the one-block form that runs the last partial round of workgroups
    for nb, min_gap in ((2, 60), (1, 30)):
        start = next(i for i, l in enumerate(all_lines) if l.startswith(f"_Z16attn_pipe_kernelILi{nb}E") and ":" in l)
        end = next(i for i in range(start, len(all_lines)) if "s_endpgm" in all_lines[i])
        lines = [l for l in all_lines[start:end + 1] if l and not l.startswith(";") and not l.startswith(".")]
        bars = [i for i, l in enumerate(lines) if l.startswith("s_barrier")]
        assert len(bars) >= 9, f"prologue + 4 main-loop steps + 4 tail steps expected, found {len(bars)} barriers"
        # barriers: [prologue, main step 0, 1, 2, 3, tail step 0 ...]; a main-loop step = from its barrier to the next one
        for k in (1, 2, 3):
            st = lines[bars[k]:bars[k + 1]]
            cnt = lambda pat: sum(1 for l in st if l.startswith(pat))
            assert cnt("v_mfma_f32_32x32x16_bf16") == 16 * nb, (nb, k, cnt("v_mfma_f32_32x32x16_bf16"))
            assert cnt("v_exp_f32") == 32 * nb and cnt("v_add_f32") == 32 * nb and cnt("v_cvt_pk_bf16_f32") == 16 * nb, \
                (nb, k, cnt("v_exp_f32"), cnt("v_add_f32"), cnt("v_cvt_pk_bf16_f32"))
            assert sum(1 for l in st if l.startswith("ds_read_b128 a[")) == 16, "16 fragment reads per step, straight into AGPRs"
            assert sum(1 for l in st if l.startswith("buffer_load_dwordx4") and l.endswith("lds")) == 4
            assert not any(l.startswith("v_accvgpr") for l in st), "accumulator moves inside a step"
            assert not any(l.startswith("v_pk_") for l in st), "packed f32 VALU inside a step"
            assert not any("scratch_" in l for l in st)
            # S chains: an MFMA with a VGPR destination; the LAST one writing a given destination, and the first later non-MFMA reader of it
            last = {}
            for i, l in enumerate(st):
                if l.startswith("v_mfma_f32_32x32x16_bf16 v["):
                    last[l.split()[1].rstrip(",")] = i
            assert len(last) == 2 * nb, last
            full = lines[bars[k]:bars[k + 2]]                      # the readers of the last chains sit in the next step
            for dst, i in last.items():
                regs = _regs(dst)
                for m in range(i + 1, len(full)):
                    t = full[m]
                    if t.startswith("v_mfma"):
                        continue
                    if any(_regs(x) & regs for x in re.findall(r"v\[\d+:\d+\]|v\d+", t)):
                        assert m - i >= min_gap, f"NB {nb} step {k}: `{t}` reads {dst} only {m - i} instructions behind the chain's last MFMA"
                        break
                else:
                    raise AssertionError(f"no reader of {dst} found")
        # ---- the ragged-tail steps: `mask_tail` WRITES the S tuples of tile j + 1 with v_cndmask right behind their chains - a WAW on
        # registers an asm MFMA is still producing, which hipcc does not model.  A 32x32x16 MFMA occupies the pipe for 16 passes, so two other
        # MFMAs (or 19 instructions) between the chain's last MFMA and the first VALU write cover the 18 wait states.  And everywhere: the
        # instruction right behind an asm v_exp_f32 must not read its destination (the trans-use wait state) ----
        n_tail_checked = 0
        for k in range(5, len(bars) - 1):
            st = lines[bars[k]:bars[k + 1]]
            last = {}
            for i, l in enumerate(st):
                if l.startswith("v_mfma_f32_32x32x16_bf16 v["):
                    last[l.split()[1].rstrip(",")] = i
            for dst, i in last.items():
                regs = _regs(dst)
                for m in range(i + 1, len(st)):
                    t = st[m]
                    if t.startswith("v_mfma") or not t.startswith("v_"):
                        continue
                    ops_ = re.findall(r"v\[\d+:\d+\]|v\d+", t)
                    if ops_ and (_regs(ops_[0]) & regs):                 # first operand = destination
                        between = sum(1 for x in st[i + 1:m] if x.startswith("v_mfma"))
                        assert between >= 2 or m - i >= 19, f"NB {nb} tail step {k}: `{t}` overwrites {dst} {m - i} instructions / {between} MFMAs behind its chain"
                        n_tail_checked += 1
                        break
        assert n_tail_checked >= 2 * nb, f"NB {nb}: the masked tail steps were not found ({n_tail_checked} chain -> VALU-write pairs)"
        for k in range(1, len(bars) - 1):
            st = lines[bars[k]:bars[k + 1]]
            for i, l in enumerate(st[:-1]):
                if l.startswith("v_exp_f32"):
                    d = _regs(l.split()[1].rstrip(","))
                    nxt = st[i + 1]
                    srcs = re.findall(r"v\[\d+:\d+\]|v\d+", nxt)
                    reads = srcs[1:] if nxt.startswith("v_") and not nxt.startswith("v_mfma") else srcs
                    assert not any(_regs(x) & d for x in reads), f"NB {nb} step {k}: `{nxt}` reads the result of `{l}` in the very next slot"
