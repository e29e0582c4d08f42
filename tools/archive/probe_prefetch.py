"""Host and device time of FrozenSources.prefetch at an emulated world size.  Usage: python tools/probe_prefetch.py WORLD GROUPS"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
W, G = int(sys.argv[1]), int(sys.argv[2])
os.environ["PLEAS_EMULATE_WORLD"] = str(W)
import torch
from pleas_merging_amd import resnet as zoo
from pleas_merging_amd.methods.pleas_merging import FrozenSources
dev = torch.device("cuda")
torch.manual_seed(0); m1 = zoo.resnet101().to(dev).eval(); m2 = zoo.resnet101().to(dev).eval()
xs = [torch.randn(16, 3, 224, 224, device=dev) for _ in range(2 * W * G)]
torch.cuda.synchronize()
for rep in range(3):
    src = FrozenSources(m1, m2, data_parallel=W > 1)
    t0 = time.time(); n = src.prefetch(xs, max_groups=G, memory_fraction=0.8); t1 = time.time(); torch.cuda.synchronize(); t2 = time.time()
    print("world %d: prefetch of %d batches (%d groups): host %.3f s, until done %.3f s, reserved %.1f GB"
          % (W, n, G, t1 - t0, t2 - t0, torch.cuda.memory_reserved() / 1e9))
    src.close(); del src
