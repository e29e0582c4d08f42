import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pleas_merging_amd import resnet as zoo
dev = torch.device("cuda")
torch.manual_seed(0); m = zoo.resnet101().to(dev).eval()
x = torch.randn(16, 3, 224, 224, device=dev)
def run(tag, mm, xx):
    with torch.no_grad():
        t0 = time.time(); mm(xx); torch.cuda.synchronize(); t1 = time.time()
        for _ in range(3): mm(xx)
        torch.cuda.synchronize(); t2 = time.time()
        for _ in range(10): mm(xx)
        torch.cuda.synchronize(); t3 = time.time()
    print("%-28s first %.2fs  steady %.2f ms/fwd" % (tag, t1 - t0, (t3 - t2) / 10 * 1e3), flush=True)
run("default", m, x)
torch.backends.cudnn.benchmark = True
run("cudnn.benchmark=True", m, x)
torch.backends.cudnn.benchmark = False
m2 = m.to(memory_format=torch.channels_last); x2 = x.to(memory_format=torch.channels_last)
run("channels_last", m2, x2)
torch.backends.cudnn.benchmark = True
run("channels_last+benchmark", m2, x2)
