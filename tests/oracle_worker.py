"""Child process of the real-size oracle comparisons (tests/test_prodshape_gpu.py): runs ONE oracle/vae.py stage on the host cores and saves the
result, so that the fp32 oracle (parent) and the bf16-emulated oracle (this process) of a 9 x 720 x 1280 frame-batch run side by side instead of
one after the other.  Test infrastructure only; no GPU.
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
    torch.set_num_threads(threads)
    v, _t, _s = config.default_configs()
    wv = weights.random_state_dict(weights.vae_param_shapes(v), seed)
    for k in ("decoder.conv_out.conv.weight", "decoder.conv_out.conv.bias"):
        wv[k] = wv[k] * scale
    x = torch.load(src)
    vae = OracleVAE(v, wv, dtype)
    out = vae.encode(x) if stage == "enc" else vae.decode(x)
    torch.save(out.float(), dst)


if __name__ == "__main__":
    main()
