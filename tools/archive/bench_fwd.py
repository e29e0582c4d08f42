"""Per-layer-class microbenchmark of pleas_fwd_batch on ResNet-101 merged-layer shapes (batch 16)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pleas_merging_amd import hip_ops
N = 16
classes = [(23, 256, 256, 14, 3, 1), (23, 1024, 256, 14, 1, 1), (22, 256, 1024, 14, 1, 1), (3, 64, 64, 56, 3, 1), (3, 256, 64, 56, 1, 1),
           (2, 64, 256, 56, 1, 1), (3, 128, 128, 28, 3, 1), (4, 512, 128, 28, 1, 1), (3, 128, 512, 28, 1, 1), (2, 512, 512, 7, 3, 1),
           (3, 2048, 512, 7, 1, 1), (2, 512, 2048, 7, 1, 1), (1, 512, 512, 14, 3, 2), (1, 2048, 1024, 14, 1, 2), (1, 64, 3, 224, 7, 2)]
for cnt, Cout, Cin, H, k, s in classes:
    pad = k // 2
    Ho = (H + 2 * pad - k) // s + 1
    batch = hip_ops.FwdBatch(torch.device("cuda"))
    r1 = torch.arange(Cout, dtype=torch.int32, device="cuda"); r2 = torch.arange(Cout, dtype=torch.int32, device="cuda")
    tens = [(torch.randn(N, Cin, H, H, device="cuda"), torch.randn(Cout, Cin, k, k, device="cuda"), torch.randn(N, Cout, Ho, Ho, device="cuda"),
             torch.randn(N, Cout, Ho, Ho, device="cuda"), torch.empty(N, Cout, Ho, Ho, device="cuda")) for _ in range(cnt)]
    loss = torch.zeros(cnt, device="cuda")
    def run():
        for ip, w, o1, o2, res in tens: batch.add(ip, w, None, o1, o2, r1, r2, Cout, res, 1.0, 1.0, (k, k), s, pad)
        batch.flush(loss)
    for _ in range(2): run()
    hip_ops.profile_reset(); hip_ops.profile_enable(True)
    reps = 10
    for _ in range(reps): run()
    torch.cuda.synchronize(); hip_ops.profile_enable(False)
    p = hip_ops.profile_collect()["conv_fwd"]
    t = p[1] / reps; fl = p[2] / reps
    print("x%2d Cout=%4d Cin=%4d H=%3d k=%d s=%d : %7.1f us  %6.1f GFLOP  %5.1f TF/s" % (cnt, Cout, Cin, H, k, s, t * 1e3, fl / 1e9, fl / t / 1e9))
