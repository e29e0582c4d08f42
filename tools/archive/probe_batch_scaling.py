"""Do the frozen source forwards get cheaper per sample at a larger batch?  Both fused ResNet-101 sources on two streams
(the PLeaS loop's arrangement), batch 16 / 32 / 48 / 64."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pleas_merging_amd import resnet as zoo
from pleas_merging_amd.methods.source_forward import fuse_bn_act
dev = torch.device("cuda")
torch.manual_seed(0); g1 = fuse_bn_act(zoo.resnet101().to(dev).eval())
torch.manual_seed(1); g2 = fuse_bn_act(zoo.resnet101().to(dev).eval())
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
work = torch.cuda.Stream()
def both(x):
    main = torch.cuda.current_stream()
    for s, g in ((s1, g1), (s2, g2)):
        s.wait_stream(main)
        with torch.cuda.stream(s):
            g(x)
    for s in (s1, s2): main.wait_stream(s)
with torch.no_grad(), torch.cuda.stream(work):
    for B in (16, 32, 48, 64, 16):
        x = torch.randn(B, 3, 224, 224, device=dev)
        for _ in range(3): both(x)
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(10): both(x)
        torch.cuda.synchronize(); dt = (time.time() - t0) / 10
        print("batch %2d: %.2f ms per pair of forwards, %.3f ms per sample" % (B, dt * 1e3, dt * 1e3 / B), flush=True)
