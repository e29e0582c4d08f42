"""Host enqueue cost of one source forward (ResNet-101, batch 16): default vs cudnn.benchmark, fused BN graph."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pleas_merging_amd import resnet as zoo
from pleas_merging_amd.methods.source_forward import fuse_bn_act
dev = torch.device("cuda")
torch.manual_seed(0); m = zoo.resnet101().to(dev).eval()
gm = fuse_bn_act(m)
x = torch.randn(16, 3, 224, 224, device=dev)
def run(tag, mm):
    with torch.no_grad():
        t0 = time.time(); mm(x); torch.cuda.synchronize(); first = time.time() - t0
        for _ in range(3): mm(x)
        torch.cuda.synchronize(); t2 = time.time(); enq = 0.0
        for _ in range(20):
            a = time.time(); mm(x); enq += time.time() - a
        torch.cuda.synchronize(); t3 = time.time()
    print("%-34s first %.2fs  enqueue %.2f ms  wall %.2f ms per forward" % (tag, first, enq / 20 * 1e3, (t3 - t2) / 20 * 1e3), flush=True)
run("modules, default", m)
run("fused bn_act graph, default", gm)
torch.backends.cudnn.benchmark = True
run("fused bn_act graph, benchmark=True", gm)
run("modules, benchmark=True", m)
