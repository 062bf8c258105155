"""What the last tile column / row of a conv costs: conv3x3_halo4x on images that end ON a 16 x 32 tile boundary, within the first half of the
next tile (the partial-tile launches of round 6: 32 x 16 tiles for that column, 8 x 64 for that row) and at the end of that next tile (what the
one-launch form paid for the partial case).  Prints ms per call (median of N) and the cost of the extra column / row relative to one tile column / row."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dove_amd import ops

BF = torch.bfloat16
dev = torch.device("cuda", 0)


def t_conv(cin, cout, T, nb, H, W, reps=7):
    g = torch.Generator(device=dev).manual_seed(1)
    pc = ops.pack_conv(torch.randn(cout, cin, 3, 3, 3) * (cin * 27) ** -0.5, torch.zeros(cout), dev)
    x = torch.randn(nb * T, H, W, cin, device=dev, generator=g).to(BF)
    cache = torch.randn(*((nb, 2) if nb > 1 else (2,)), H, W, cin, device=dev, generator=g).to(BF)
    for _ in range(2):
        ops.conv(x, pc, cache=cache, nb=nb)
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.conv(x, pc, cache=cache, nb=nb); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


for name, cin, cout, T, nb, shapes, units in (
        ("128->128, 12 tiles x 8 frames, 240 rows: width", 128, 128, 8, 12, [(240, 352), (240, 360), (240, 384)], 11),
        ("256->256, 12 tiles x 4 frames, 180 cols: height", 256, 256, 4, 12, [(112, 192), (120, 192), (128, 192)], 7),
        ("256->256, 1 x 4 frames, 640 cols: height (the 360-row level of the 720p clip)", 256, 256, 4, 1, [(352, 640), (360, 640), (368, 640)], 22),
        ("512->512, 12 tiles x 2 frames, 30 rows: width", 512, 512, 2, 12, [(30, 32), (30, 45), (30, 64)], 1)):
    ms = [t_conv(cin, cout, T, nb, H, W) for H, W in shapes]
    unit = ms[0] / units
    print(f"{name}: " + "  ".join(f"{H}x{W}: {m:.3f} ms" for (H, W), m in zip(shapes, ms)) +
          f"   | one full tile column/row = {unit:.3f} ms; the partial one costs {(ms[1] - ms[0]) / unit:.2f} of it, a full extra one {(ms[2] - ms[0]) / unit:.2f}")
