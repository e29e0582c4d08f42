"""bn_act / bn_act_tracked standalone on the shapes of a 128-sample ResNet-101 source forward: effective HBM GB/s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pleas_merging_amd import hip_ops
dev = torch.device("cuda")
shapes = [(128, 64, 112, 112), (128, 64, 56, 56), (128, 256, 56, 56), (128, 128, 28, 28), (128, 512, 28, 28), (128, 256, 14, 14),
          (128, 1024, 14, 14), (128, 512, 7, 7), (128, 2048, 7, 7), (16, 256, 14, 14), (16, 1024, 14, 14), (64, 1024, 14, 14)]
for shp in shapes:
    x = torch.randn(shp, device=dev); r = torch.randn(shp, device=dev)
    s = torch.rand(shp[1], device=dev) + 0.5; t = torch.randn(shp[1], device=dev)
    for name, fn, passes in (("bn+relu", lambda: hip_ops.bn_act(x, s, t, None, True), 2), ("bn+add+relu", lambda: hip_ops.bn_act(x, s, t, r, True), 3),
                             ("tracked bn+add+relu (3 outputs)", lambda: hip_ops.bn_act_tracked(x, s, t, r, True, True), 5)):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print("%-22s %-34s %7.1f us  %6.0f GB/s" % (str(shp), name, dt * 1e6, passes * x.numel() * 4 / dt / 1e9), flush=True)

# the stem chain bn1 -> relu -> maxpool(3, 2, 1): two passes (bn_act, vendor pooling) against the pooled pass
import torch.nn.functional as F
for shp in [(128, 64, 112, 112), (64, 64, 112, 112), (16, 64, 112, 112)]:
    x = torch.randn(shp, device=dev); s = torch.rand(shp[1], device=dev) + 0.5; t = torch.randn(shp[1], device=dev)
    for name, fn in (("bn_act, then vendor max_pool2d", lambda: F.max_pool2d(hip_ops.bn_act(x, s, t, None, True), 3, 2, 1)),
                     ("vendor max_pool2d alone", lambda: F.max_pool2d(x, 3, 2, 1)),
                     ("bn_act_maxpool", lambda: hip_ops.bn_act_maxpool(x, s, t, (3, 3), 2, 1, True))):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print("%-22s %-34s %7.1f us  %6.0f GB/s of (x + pooled y)" % (str(shp), name, dt * 1e6, 1.25 * x.numel() * 4 / dt / 1e9), flush=True)
