"""Host cost of building the twin matching graph and the fused PLeaS sources (runs without a GPU): cProfile."""
import sys, os, time, cProfile, pstats, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pleas_merging_amd import resnet as zoo, hip_ops
from pleas_merging_amd.core.compiler import get_permutation_spec
am = importlib.import_module("pleas_merging_amd.methods.activation_matching")
pm = importlib.import_module("pleas_merging_amd.methods.pleas_merging")
torch.manual_seed(0); m1 = zoo.resnet101().eval(); m2 = zoo.resnet101().eval()
spec = get_permutation_spec(m1, ((1, 3, 224, 224),))
dev = torch.device("cuda" if torch.cuda.is_available() else "cpu")
if dev.type == "cuda":
    m1.to(dev); m2.to(dev)
def twin():
    arena = am.GroupArena(spec, dev)
    return am.build_fused_module(spec, m1, m2, arena, hip_ops.EPI_NEG_CDIST, True, overlap=True, fuse_bn=True, derive_bn=True)
for name, fn in (("twin graph", twin), ("prepare_sources", lambda: pm.prepare_sources(m1, m2))):
    try:
        fn()
    except Exception as e:
        print(name, "failed on this host:", type(e).__name__, e); continue
    t0 = time.time(); fn(); print("%s: %.3f s" % (name, time.time() - t0))
    pr = cProfile.Profile(); pr.enable(); fn(); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
if dev.type == "cuda":   # the whole matching call on a few batches: what runs before the first batch is requested?
    xs = [(torch.randn(16, 3, 224, 224, device=dev), None) for _ in range(3)]
    am.activation_matching(spec, m1, m2, xs, 3, output_costs=True)
    torch.cuda.synchronize()
    marks = []
    def gen():
        for i, b in enumerate(xs):
            torch.cuda.synchronize(); marks.append(time.time()); yield b
    pr = cProfile.Profile(); t0 = time.time(); pr.enable()
    am.activation_matching(spec, m1, m2, gen(), 3, output_costs=True)
    pr.disable(); torch.cuda.synchronize()
    print("matching call: first batch requested after %.3f s; batches %.3f s; total %.3f s" % (marks[0] - t0, marks[-1] - marks[0], time.time() - t0))
    pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
