"""Per-layer-class microbenchmark of pleas_wgrad_batch on ResNet-101 merged-layer shapes (batch 16)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pleas_merging_amd import hip_ops
N = 16
# (count, Cout, Cin, H, k, stride)  -- H = input size
classes = [(23, 256, 256, 14, 3, 1), (23, 1024, 256, 14, 1, 1), (22, 256, 1024, 14, 1, 1), (3, 64, 64, 56, 3, 1), (3, 256, 64, 56, 1, 1),
           (2, 64, 256, 56, 1, 1), (3, 128, 128, 28, 3, 1), (4, 512, 128, 28, 1, 1), (3, 128, 512, 28, 1, 1), (2, 512, 512, 7, 3, 1),
           (3, 2048, 512, 7, 1, 1), (2, 512, 2048, 7, 1, 1), (1, 512, 512, 14, 3, 2), (1, 2048, 1024, 14, 1, 2)]
tot_t = tot_f = 0
for cnt, Cout, Cin, H, k, s in classes:
    pad = k // 2
    Ho = (H + 2 * pad - k) // s + 1
    batch = hip_ops.WgradBatch(torch.device("cuda"))
    tens = []
    for _ in range(cnt):
        tens.append((torch.randn(N, Cout, Ho, Ho, device="cuda"), torch.randn(N, Cin, H, H, device="cuda"), torch.empty(Cout, Cin, k, k, device="cuda")))
    def run():
        for r, i, g in tens: batch.add(r, i, g, (k, k), s, pad)
        batch.flush()
    for _ in range(2): run()
    hip_ops.profile_reset(); hip_ops.profile_enable(True)
    reps = 10
    for _ in range(reps): run()
    torch.cuda.synchronize(); hip_ops.profile_enable(False)
    p = hip_ops.profile_collect()["conv_wgrad"]
    t = p[1] / reps; fl = p[2] / reps
    print("x%2d Cout=%4d Cin=%4d H=%3d k=%d s=%d : %7.1f us  %6.1f GFLOP  %5.1f TF/s" % (cnt, Cout, Cin, H, k, s, t * 1e3, fl / 1e9, fl / t / 1e9))
    tot_t += t; tot_f += fl
print("sum of classes: %.2f ms, %.1f GFLOP -> %.1f TF/s (launched per class, not as one grid)" % (tot_t, tot_f / 1e9, tot_f / tot_t / 1e9))
