"""Per-shape microbenchmark of pleas_gram_accum on the ResNet-101 node shapes (GPU box only)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pleas_merging_amd import hip_ops, _lib
B = 16
shapes = [(135, 256, 196), (94, 1024, 196), (19, 64, 3136), (21, 128, 784), (18, 512, 784), (15, 512, 49), (14, 2048, 49),
          (14, 256, 3136), (3, 64, 12544), (2, 2048, 1)]
tunes = [(512, 2)] + [tuple(map(int, a.split(","))) for a in sys.argv[1:]]
for tb, mc in tunes:
    _lib.lib().pleas_gram_tune(tb, mc)
    tot_t = tot_f = 0.0
    print("== target_blocks=%d min_chunks=%d" % (tb, mc))
    for count, C, HW in shapes:
        x = torch.randn(B, C, HW, device="cuda"); y = torch.randn(B, C, HW, device="cuda")
        acc = torch.zeros(C, C, device="cuda")
        for _ in range(3): hip_ops.gram_accum(x, y, 1, acc, hip_ops.EPI_NEG_CDIST, True)
        hip_ops.profile_reset(); hip_ops.profile_enable(True)
        reps = 20
        for _ in range(reps): hip_ops.gram_accum(x, y, 1, acc, hip_ops.EPI_NEG_CDIST, True)
        torch.cuda.synchronize(); hip_ops.profile_enable(False)
        p = hip_ops.profile_collect()
        tp, tf = p["gram_partial"][1] / reps, p["gram_finalize"][1] / reps
        fl = 2.0 * C * C * B * HW
        print("C=%4d HW=%5d x%3d  partial %7.1f us (%5.1f TF/s)  finalize %6.1f us  ws=%5.1f MB  total/batch %.2f ms" %
              (C, HW, count, tp * 1e3, fl / tp / 1e9, tf * 1e3, hip_ops.gram_ws_bytes(B, C, HW) / 2**20, count * (tp + tf)))
        tot_t += count * (tp + tf); tot_f += count * fl
    print("   per batch: %.2f ms for %.1f GFLOP -> %.1f TF/s overall" % (tot_t, tot_f / 1e9, tot_f / tot_t / 1e9))
