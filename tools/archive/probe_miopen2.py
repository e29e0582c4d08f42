"""ResNet-101 forward (batch 16, fp32): MIOpen immediate mode (PyTorch default) vs Find mode (cudnn.benchmark=True set
before the first convolution), fresh process per mode: python tools/probe_miopen2.py {0|1}"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.backends.cudnn.benchmark = bool(int(sys.argv[1]))
from pleas_merging_amd import resnet as zoo
dev = torch.device("cuda")
torch.manual_seed(0); m = zoo.resnet101().to(dev).eval()
x = torch.randn(16, 3, 224, 224, device=dev)
with torch.no_grad():
    t0 = time.time(); m(x); torch.cuda.synchronize(); first = time.time() - t0
    for _ in range(3): m(x)
    torch.cuda.synchronize(); t2 = time.time()
    for _ in range(20): m(x)
    torch.cuda.synchronize(); t3 = time.time()
print("benchmark=%s: first forward %.2f s, steady %.2f ms per forward" % (torch.backends.cudnn.benchmark, first, (t3 - t2) / 20 * 1e3))
