"""Cost of capturing both fused source forwards (two side streams) into one hipGraph, and replay time."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pleas_merging_amd import resnet as zoo
from pleas_merging_amd.methods.source_forward import fuse_bn_act
dev = torch.device("cuda")
torch.manual_seed(0); g1 = fuse_bn_act(zoo.resnet101().to(dev).eval())
torch.manual_seed(1); g2 = fuse_bn_act(zoo.resnet101().to(dev).eval())
x = torch.randn(16, 3, 224, 224, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def both():
    main = torch.cuda.current_stream()
    outs = []
    for s, g in ((s1, g1), (s2, g2)):
        s.wait_stream(main)
        with torch.cuda.stream(s):
            outs.append(g(x))
    for s in (s1, s2): main.wait_stream(s)
    return outs
with torch.no_grad():
    for _ in range(3): both()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(10): both()
    torch.cuda.synchronize(); print("eager two-stream: %.2f ms per pair" % ((time.time() - t0) / 10 * 1e3))
    graph = torch.cuda.CUDAGraph()
    torch.cuda.synchronize(); t0 = time.time()
    with torch.cuda.graph(graph):
        outs = both()
    t1 = time.time(); torch.cuda.synchronize(); t2 = time.time()
    graph.replay(); torch.cuda.synchronize(); t3 = time.time()
    print("capture (incl. instantiate) %.3f s, sync %.3f s, first replay %.3f s" % (t1 - t0, t2 - t1, t3 - t2))
    torch.cuda.synchronize(); t0 = time.time(); enq = 0
    for _ in range(20):
        a = time.time(); graph.replay(); enq += time.time() - a
    torch.cuda.synchronize(); print("replay: enqueue %.3f ms, wall %.2f ms per pair" % (enq / 20 * 1e3, (time.time() - t0) / 20 * 1e3))
# ---- same, with every conv input/output kept alive (the fitter's taps): the graph's private pool must hold ~7 GB
keep = []
hooks = [m.register_forward_hook(lambda mod, i, o: keep.append((i[0], o))) for g in (g1, g2) for m in g.modules() if isinstance(m, torch.nn.Conv2d)]
with torch.no_grad():
    both(); torch.cuda.synchronize(); keep.clear()
    t0 = time.time(); a = torch.empty(7 * 1024**3, dtype=torch.uint8, device=dev); torch.cuda.synchronize(); t1 = time.time()
    print("fresh 7 GiB allocation through the caching allocator: %.3f s" % (t1 - t0)); del a
    graph2 = torch.cuda.CUDAGraph()
    torch.cuda.synchronize(); t0 = time.time()
    with torch.cuda.graph(graph2):
        outs = both()
    t1 = time.time(); torch.cuda.synchronize()
    graph2.replay(); torch.cuda.synchronize(); t3 = time.time()
    print("capture with %d live taps (%.1f GB): %.3f s, first replay %.3f s" % (len(keep), sum(o.numel() for _, o in keep) * 4 / 1e9 , t1 - t0, t3 - t1))
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(20): graph2.replay()
    torch.cuda.synchronize(); print("replay with live taps: wall %.2f ms per pair" % ((time.time() - t0) / 20 * 1e3))
    print(torch.cuda.memory_reserved() / 1e9, "GB reserved")
