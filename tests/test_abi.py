"""The C-ABI library loads and exports every symbol include/dove_hip.h declares (no compute calls: CPU-safe)."""
import os
import re

import pytest

from dove_amd import lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(L.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return L.load()


def header_symbols():
    with open(os.path.join(ROOT, "include", "dove_hip.h")) as f:
        src = f.read()
    return sorted(set(re.findall(r"\b(dove_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(lib):
    syms = header_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/dove_hip.h but not exported"


def test_binding_covers_header():
    assert set(header_symbols()) == set(L.SIGNATURES) | set(L.PLAIN)


def test_version_and_error_channel(lib):
    assert lib.dove_abi_version() == 15
    # argument validation happens before any HIP call, so it is safe without a GPU
    rc = lib.dove_axpby(None, None, None, 0, 0, 1.0, 1.0, None)
    assert rc == -1 and b"axpby" in lib.dove_last_error()


def test_conv_desc_struct_size_is_checked(lib):
    """dove_conv_desc is caller-allocated and grows with the ABI: every entry point that takes one refuses a struct_size other than
    its own sizeof (a binding written against an older header would have its missing tail fields read from stray memory)."""
    import ctypes as C
    d = L.ConvDesc()
    assert d.struct_size == C.sizeof(L.ConvDesc) and C.sizeof(L.ConvDesc) % 8 == 0
    d.x = d.w = d.out = 1
    d.struct_size -= 8                                           # what a pre-ABI-8 binding (no out_f32) would have passed
    assert lib.dove_conv_igemm_bf16(C.byref(d), None) == -1 and b"struct_size" in lib.dove_last_error()
    assert lib.dove_conv_kernel_name(C.byref(d)) == b"" and int(lib.dove_conv_gn_partial_rows(C.byref(d))) == 0
    d.struct_size = 0
    assert lib.dove_conv_igemm_bf16(C.byref(d), None) == -1
    assert not hasattr(L.ConvDesc, "debug_buf") and not hasattr(lib, "dove_timing_set_debug_buf")      # timing hook: timing library only


def test_dispatch_rule_mirror(lib):
    """`dove_conv_gn_partial_rows` is pure host logic (no HIP call): it answers "does this conv dispatch to the kernel that
    fuses the GroupNorm statistics, and how many partial rows will it write" - from the same selection rule that
    `dove_conv_kernel_name` reports, on every shape class of the VAE."""
    import ctypes as C

    import torch

    from dove_amd import ops

    def desc(cin, cout, k, T, H, W, up=0, stride=1):
        pc = ops.PackedConv(torch.empty(0), None, *( (1,) + k if len(k) == 2 else k), cin, (cin + 31) // 32 * 32, cout,
                            (cout + 31) // 32 * 32)
        d = L.ConvDesc()
        d.x = d.w = d.out = 1                                   # never dereferenced by the rows query
        d.t_in, d.h_in, d.w_in, d.cin = T, H, W, pc.cin_pad
        d.t_out, d.h_out, d.w_out = T, H << up, W << up
        if stride == 2:
            d.h_out, d.w_out = (H + 1 - pc.kh) // 2 + 1, (W + 1 - pc.kw) // 2 + 1
        d.cout_pad, d.cout_store = pc.cout_pad, pc.cout_store
        d.kt, d.kh, d.kw, d.stride = pc.kt, pc.kh, pc.kw, stride
        d.pad_h, d.pad_w = (0, 0) if stride == 2 else (1 if pc.kh == 3 else 0, 1 if pc.kw == 3 else 0)
        d.up, d.tmode, d.act = up, 0, 0
        d.ldo = pc.cout_store
        return pc, d

    cases = [  # cin, cout, k, T, H, W, up, stride, fused?
        (128, 128, (3, 3, 3), 8, 720, 1280, 0, 1, True),     # encoder / decoder L0 resnet convs
        (256, 256, (3, 3, 3), 4, 360, 640, 0, 1, True),
        (512, 512, (3, 3, 3), 2, 90, 160, 0, 1, True),
        (256, 256, (3, 3), 4, 360, 640, 1, 1, True),         # upsample-fused conv
        (128, 128, (3, 3), 8, 720, 1280, 0, 2, False),       # stride-2 downsample: generic kernel, separate stats pass
        (32, 128, (3, 3, 3), 9, 720, 1280, 0, 1, False),     # direct conv_in (Cin padded 3 -> 32; tiled path only): generic kernel
        (128, 32, (3, 3, 3), 8, 720, 1280, 0, 1, False),     # conv_out (Cout 3 -> 32)
        (512, 512, (3, 3, 3), 2, 8, 32, 0, 1, False),        # H < 16: 4-wave halo kernel
        (3072, 9216, (1, 1, 1), 1, 1, 18226, 0, 1, False),   # a DiT linear
    ]
    for cin, cout, k, T, H, W, up, stride, fused in cases:
        pc, d = desc(cin, cout, k, T, H, W, up, stride)
        rows = int(lib.dove_conv_gn_partial_rows(C.byref(d)))
        ph = pw = 1 if stride == 1 and pc.kh == 3 else 0
        var = lib.dove_conv_kernel_name(C.byref(d)).decode()
        assert (rows > 0) == fused, (cin, cout, k, H, W, rows)
        assert (var == "conv3x3_halo4x_kernel") == fused, (cin, cout, k, H, W, var)
        if fused:
            assert rows == d.t_out * -(-d.h_out // 16) * -(-d.w_out // 32) * 4
    names = {  # production shape -> kernel (cin, cout, k, T, H, W, up, stride)
        (32, 1024, (1, 1, 1), 2, 90, 160, 0, 1): "smallk_kernel",           # SpatialNorm conv_y || conv_b on the latent grid
        (3072, 9216, (1, 1, 1), 1, 1, 18226, 0, 1): "gemm8p_kernel",
        (32, 128, (3, 3, 3), 9, 720, 1280, 0, 1): "igemm_fast_kernel",
        (128, 32, (3, 3, 3), 8, 720, 1280, 0, 1): "igemm_fast_kernel",
        (128, 128, (3, 3), 8, 720, 1280, 0, 2): "igemm_fast_kernel",
    }
    for key, want in names.items():
        pc, d = desc(*key)
        assert lib.dove_conv_kernel_name(C.byref(d)).decode() == want, (key, want)


def test_graft_entry_build_version_check(lib):
    """__graft_entry__.build() checks the loaded library against the header's DOVE_ABI_VERSION (not a literal)."""
    import inspect

    import __graft_entry__ as g
    src = inspect.getsource(g.build)
    assert "DOVE_ABI_VERSION" in src and "== 1" not in src
    with open(os.path.join(ROOT, "include", "dove_hip.h")) as f:
        want = int(re.search(r"#define\s+DOVE_ABI_VERSION\s+(\d+)", f.read()).group(1))
    assert lib.dove_abi_version() == want



def test_product_library_has_no_work_skipping_switches():
    """ablation switches (skip A/B loads, skip MFMA) and s_memtime timing instantiations live only in the
    separate -DDOVE_TIMING_BUILD library; the product .so must not even contain the environment variable's name."""
    import os
    from dove_amd import lib as L
    blob = open(L.LIB_PATH, "rb").read()
    assert b"DOVE_IGEMM_ABLATE" not in blob
    assert not os.path.basename(L.LIB_PATH).endswith("_timing.so")


def test_kernel_dispatch_table():
    """dove_conv_kernel_name is the library's ONE selection rule (no environment switches).  Pinned here: (i) the production
    shapes of the 33x720x1280 clip run on the kernels DESIGN.md names for them; (ii) every kernel of the library is reached by
    at least one GPU parity case of tests/test_ops_gpu.py (so no kernel depends on a fallback re-run for its coverage)."""
    import math

    import torch

    import test_ops_gpu as T
    from dove_amd import ops

    def pc(cout, cin, k):
        return ops.pack_conv(torch.zeros(cout, cin, *k), None, "cpu")

    def name(x_shape, cout, cin, k, **kw):
        p = pc(cout, cin, k)
        return ops.conv_kernel_name(x_shape[:3] + (p.cin_pad,), p, **kw)

    prod = {
        "resnet conv 128->128 @ 9x720x1280": (name((9, 720, 1280), 128, 128, (3, 3, 3), resid=True), "conv3x3_halo4x_kernel"),
        "resnet conv 512->512 @ 3x90x160": (name((3, 90, 160), 512, 512, (3, 3, 3)), "conv3x3_halo4x_kernel"),
        "upsample conv 256->256 @ 360x640 -> 720x1280": (name((8, 360, 640), 256, 256, (3, 3), up=1, pad=(1, 1)), "conv3x3_halo4x_kernel"),
        "encoder.conv_in 3->128 (direct form: tiled VAE only)": (name((9, 720, 1280), 128, 3, (3, 3, 3)), "igemm_fast_kernel"),
        "decoder.conv_in 16->512": (name((3, 90, 160), 512, 16, (3, 3, 3)), "igemm_fast_kernel"),
        "decoder.conv_out 128->3": (name((9, 720, 1280), 3, 128, (3, 3, 3)), "igemm_fast_kernel"),
        "downsample conv stride 2": (name((9, 720, 1280), 128, 128, (3, 3), stride=2, pad=(0, 0)), "igemm_fast_kernel"),
        "DiT qkv 3072->9216": (name((1, 1, 18226), 9216, 3072, ()), "gemm8p_kernel"),
        "DiT ff2 12288->3072 gated": (name((1, 1, 18226), 3072, 12288, (), gated=True), "gemm8p_kernel"),
        "SpatialNorm conv_y||conv_b 16->256": (name((3, 90, 160), 256, 16, (1, 1, 1)), "smallk_kernel"),
        "resnet shortcut 256->128 @ 720p": (name((9, 720, 1280), 128, 256, (1, 1, 1)), "igemm_fast_kernel"),
        "text embedding 4096->3072 (226 rows)": (name((1, 1, 226), 3072, 4096, ()), "igemm_fast_kernel"),
    }
    for what, (got, want) in prod.items():
        assert got == want, (what, got, want)
    covered = set()
    for cname, cin, cout, k, Tt, H, W, kw in T.CONV_CASES:
        kw = {a: b for a, b in kw.items() if a not in ("cache",)}
        resid = kw.pop("resid", False)
        covered.add(name((Tt, H, W), cout, cin, k, resid=resid, **kw))
    for cname, N, cin, cout, kw in T.LIN_CASES:
        covered.add(name((1, 1, N), cout, cin, (), act=kw.get("act", 0), gated=kw.get("gate", False), resid=kw.get("resid_only", False)))
    # the superseded 8-wave generations (conv3x3_halo8, gemm8) no longer exist: five kernels carry every shape
    assert covered == {"igemm_kernel", "igemm_fast_kernel", "conv3x3_halo4x_kernel", "gemm8p_kernel", "smallk_kernel"}, covered


def test_header_is_self_contained_c_and_links(tmp_path):
    """include/dove_hip.h is the whole contract of a non-Python host: it must compile as plain C99 on its own (round 6 found it leaning on a
    `size_t` somebody else had to declare) and a C program must link against libdove_hip.so and read the ABI version back."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "host.c"
    src.write_text('#include "dove_hip.h"\n#include <stdio.h>\n'
                   'int main(void) { dove_conv_desc d; d.struct_size = (unsigned)sizeof d; (void)d;\n'
                   '  printf("%d %d\\n", dove_abi_version(), DOVE_ABI_VERSION); return dove_abi_version() == DOVE_ABI_VERSION ? 0 : 1; }\n')
    exe = tmp_path / "host"
    lib_dir = os.path.join(root, "dove_amd")
    subprocess.check_call([gcc, "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(root, "include"), str(src), "-o", str(exe),
                           "-L", lib_dir, "-l:libdove_hip.so", f"-Wl,-rpath,{lib_dir}"])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, (out.stdout, out.stderr)
    a, b = out.stdout.split()
    assert a == b


def test_conv_partial_launch_plan():
    """dove_conv_partial_launches is a pure host function of the descriptor (no GPU): the last tile column of a conv3x3_halo4x call goes to
    a launch of its own (32 x 16 tiles) only when the image ends within the first half of it AND the two launches need fewer rounds of the
    256 persistent workgroups than one launch.  The 240 x 360 tiles of diffusers' tiled VAE (12 interior tiles in one call,
    /root/reference/inference_script.py:642-645) qualify; the 720p clip's own levels and small calls keep the one launch."""
    from dove_amd import ops

    def pc(cin, cout, kt=3):
        return ops.PackedConv(w=None, bias=None, kt=kt, kh=3, kw=3, cin=cin, cin_pad=cin, cout=cout, cout_pad=cout)

    def plan(shape, p, nb=1):
        assert ops.conv_kernel_name(shape, p, nb=nb) == "conv3x3_halo4x_kernel"
        return ops.conv_kernel_name(shape, p, nb=nb, partial=True)

    assert plan((8, 240, 360, 128), pc(128, 128), 12) == 1          # 68 rounds -> 62 + 3
    assert plan((9, 240, 360, 128), pc(128, 128), 12) == 1
    assert plan((8, 240, 360, 128), pc(128, 128), 1) == 0           # one tile at a time: 6 rounds either way
    assert plan((8, 720, 1280, 128), pc(128, 128)) == 0             # ends on tile boundaries
    assert plan((8, 360, 640, 256), pc(256, 256)) == 0
    assert plan((3, 30, 45, 512), pc(512, 512), 12) == 0            # W % 32 = 13, but one round more, not fewer
    assert plan((3, 24, 40, 128), pc(128, 128)) == 0                # test-sized: one round
    assert plan((16, 512, 40, 128), pc(128, 128, 1)) == 1
    assert plan((16, 512, 52, 128), pc(128, 128, 1)) == 0           # W % 32 = 20: the column is more than half full
    assert ops.conv_kernel_name((300, 1, 1, 3072), pc(3072, 3072, 1), partial=True) == 0     # not a halo conv at all
