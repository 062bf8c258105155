"""Child process of the real-size oracle comparisons (tests/test_prodshape_gpu.py): runs ONE oracle/vae.py stage on the host cores and saves the
result - the bf16-emulated oracle beside the parent's fp32 pass (DOVE_TEST_BF16_YARDSTICK=1), or the fp32 pass itself started by
tests/conftest.py at the beginning of a GPU session (tests/oracle_prefetch.py) so that it runs while the GPU tests do.  Test infrastructure only; no GPU.
    python tests/oracle_worker.py <enc|dec> <seed> <dtype: float32|bfloat16> <threads> <input.pt> <output.pt> [conv_out_scale]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from dove_amd import config, weights  # noqa: E402
from oracle.vae import OracleVAE  # noqa: E402


def main():
    stage, seed, dtype, threads, src, dst = sys.argv[1], int(sys.argv[2]), getattr(torch, sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6]
    scale = float(sys.argv[7]) if len(sys.argv) > 7 else 1.0
    import time
    t0 = time.time()
    torch.set_num_threads(threads)
    v, _t, _s = config.default_configs()
    wv = weights.random_state_dict(weights.vae_param_shapes(v), seed)
    for k in ("decoder.conv_out.conv.weight", "decoder.conv_out.conv.bias"):
        wv[k] = wv[k] * scale
    x = torch.load(src)
    vae = OracleVAE(v, wv, dtype)
    out = vae.encode(x.to(dtype)) if stage == "enc" else vae.decode(x.to(dtype))
    # the checksum of the input that was read travels with the result (tests/oracle_prefetch.py accepts a result only for ITS input);
    # written under another name and renamed so that a reader never sees a partial file
    xs = x.double()
    torch.save({"out": out.float(), "in_sum": float(xs.sum()) + float(xs.abs().sum()), "seconds": time.time() - t0}, dst + ".part")
    os.replace(dst + ".part", dst)


if __name__ == "__main__":
    main()
