import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dove_amd import ops
M, K, N = 4096, 256, 256
w = torch.zeros(N, K); x = torch.zeros(M, K); b = torch.arange(N).float()
pc = ops.pack_conv(w, b, "cuda")
xg = x.cuda().to(torch.bfloat16)
buf = torch.zeros(64 * 24, dtype=torch.float32, device="cuda")
y = ops.conv(xg.view(1, 1, M, K), pc, debug_buf=buf).view(M, N).float().cpu()
d = buf.cpu().view(64, 24)
for lane in (4, 8, 12, 13, 14, 15, 28, 44):
    r = d[lane]
    vb = r[16:20].view(torch.int32)
    print(lane, "lo", r[0:4].tolist(), "hi", r[4:8].tolist(), "x0", r[8:12].tolist(), "x1", r[12:16].tolist(), "v", [hex(int(t) & 0xffffffff) for t in vb], "off", int(r[20:21].view(torch.int32)))
print("y[9,96:104]", y[9, 96:104].tolist())
