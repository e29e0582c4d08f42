"""Would running both source models' layers as ONE grouped convolution (groups=2, channels concatenated) beat two
convolutions on two streams?  Main ResNet-101 layer shapes at batch 32 (the paired source forward)."""
import sys, os, time
import torch, torch.nn.functional as F
dev = torch.device("cuda"); N = 32
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def bench(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / reps * 1e6
for (ci, co, h, k) in ((256, 1024, 14, 1), (1024, 256, 14, 1), (256, 256, 14, 3), (64, 64, 56, 3), (512, 512, 7, 3), (128, 512, 28, 1)):
    xa, xb = torch.randn(N, ci, h, h, device=dev), torch.randn(N, ci, h, h, device=dev)
    wa, wb = torch.randn(co, ci, k, k, device=dev), torch.randn(co, ci, k, k, device=dev)
    xg, wg = torch.cat([xa, xb], 1), torch.cat([wa, wb], 0)
    pad = k // 2
    def two_streams():
        main = torch.cuda.current_stream()
        s1.wait_stream(main); s2.wait_stream(main)
        with torch.cuda.stream(s1): F.conv2d(xa, wa, padding=pad)
        with torch.cuda.stream(s2): F.conv2d(xb, wb, padding=pad)
        main.wait_stream(s1); main.wait_stream(s2)
    def serial():
        F.conv2d(xa, wa, padding=pad); F.conv2d(xb, wb, padding=pad)
    def grouped():
        F.conv2d(xg, wg, padding=pad, groups=2)
    with torch.no_grad():
        fl = 2 * 2.0 * N * co * ci * k * k * h * h
        a, b, c = bench(serial), bench(two_streams), bench(grouped)
        print("Cin %4d Cout %4d %2dx%-2d k%d: serial %6.1f us  two streams %6.1f us  grouped %6.1f us  (%.0f / %.0f / %.0f TF/s)" %
              (ci, co, h, h, k, a, b, c, fl / a / 1e6, fl / b / 1e6, fl / c / 1e6), flush=True)
