"""Feasibility probe: per-layer launches of the flat-shift forward tile (full epilogue: a pessimistic stand-in for a
source-forward epilogue) against the vendor convolution, ResNet-101 layer-3 shapes, batch 32 and 128."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from pleas_merging_amd import hip_ops

def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / reps * 1e-3

shapes = [(256, 256, 14, 3), (1024, 256, 14, 1), (256, 1024, 14, 1), (128, 128, 28, 3), (512, 128, 28, 1), (128, 512, 28, 1),
          (64, 64, 56, 3), (256, 64, 56, 1), (64, 256, 56, 1), (512, 512, 7, 3), (2048, 512, 7, 1), (512, 2048, 7, 1)]
for N in (32, 128):
    for Cout, Cin, H, k in shapes:
        pad = k // 2
        ip = torch.randn(N, Cin, H, H, device="cuda")
        w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.05
        o1 = torch.randn(N, Cout, H, H, device="cuda"); o2 = torch.randn(N, Cout, H, H, device="cuda")
        res = torch.empty(N, Cout, H, H, device="cuda")
        r = torch.arange(Cout, dtype=torch.int32, device="cuda")
        wk = w.permute(0, 2, 3, 1).contiguous() if k > 1 else w
        batch = hip_ops.FwdBatch(torch.device("cuda")); loss = torch.zeros(1, device="cuda")
        def own():
            batch.add(ip, wk, None, o1, o2, r, r, Cout, res, 1.0, 1.0, (k, k), 1, pad, flags=hip_ops.FwdBatch.KPOS_MAJOR if k > 1 else 0)
            batch.flush(loss)
        def vendor():
            F.conv2d(ip, w, None, 1, pad)
        fl = 2.0 * Cout * Cin * k * k * N * H * H
        t_own, t_ven = timeit(own), timeit(vendor)
        print("N=%3d Cout=%4d Cin=%4d H=%2d k=%d: own %7.1f us %6.1f TF/s | vendor %7.1f us %6.1f TF/s" %
              (N, Cout, Cin, H, k, t_own * 1e6, fl / t_own / 1e12, t_ven * 1e6, fl / t_ven / 1e12), flush=True)
