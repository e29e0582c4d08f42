"""Host/GPU time of rank 0's share of an N-rank job on ONE GPU (PLEAS_EMULATE_WORLD): phases, then cProfile of the updates.
Usage: python tools/probe_dp_rank.py WORLD [graph=1] [lookahead=1] [sources_per_forward=2]"""
import sys, os, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
GRAPH = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
LOOK = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
GROUP = int(sys.argv[4]) if len(sys.argv) > 4 else 2
os.environ["PLEAS_EMULATE_WORLD"] = str(W)
import torch
from pleas_merging_amd import resnet as zoo
from pleas_merging_amd.core.compiler import get_permutation_spec
from pleas_merging_amd.methods.activation_matching import activation_matching
from pleas_merging_amd.methods.partial_matching import partial_merge
from pleas_merging_amd.methods.pleas_merging import PleasFitter, prepare_sources
dev = torch.device("cuda"); B = 16
torch.manual_seed(0); m1 = zoo.resnet101().to(dev)
torch.manual_seed(1); m2 = zoo.resnet101().to(dev)
xs = [torch.randn(B, 3, 224, 224, device=dev) for _ in range(101)]
with torch.no_grad():
    zoo.calibrate_bn(m1, xs[:4]); zoo.calibrate_bn(m2, xs[:4])
m1.eval(); m2.eval()
spec = get_permutation_spec(m1, ((1, 3, 224, 224),))
def sync(): torch.cuda.synchronize(); return time.time()
def job(nm, nu, prof=None):
    early = {}
    t0 = sync()
    perm, costs = activation_matching(spec, m1, m2, [(x, None) for x in xs[:nm]], nm, output_costs=True,
                                      while_solving=lambda: early.update(s=prepare_sources(m1, m2)))
    t1 = sync()
    m3 = partial_merge(spec, m1, m2, perm, costs, 0.0, device=dev)
    t2 = sync()
    fit = PleasFitter(m1, m2, m3, spec, perm, costs, 0.0, nu - 1, data_parallel=True, fused_sources=early["s"])
    t3 = sync()
    if prof: prof.enable()
    h0 = time.time()
    for _ in fit.steps((xs[i % len(xs)] for i in range(nu)), lookahead=LOOK, sources_per_forward=GROUP): pass
    h1 = time.time()
    if prof: prof.disable()
    t4 = sync()
    fit.finish()
    t5 = sync()
    print("world %d graph %d lookahead %d group %d: matching+LAP %.3f  merge %.3f  init %.3f  updates %.3f (host %.3f, %.2f ms each)  finish %.3f"
          % (W, GRAPH, LOOK, GROUP, t1 - t0, t2 - t1, t3 - t2, t4 - t3, h1 - h0, (t4 - t3) * 1e3 / nu, t5 - t4))
job(W * 2, GROUP + 1)
job(100, 101)
pr = cProfile.Profile(); job(W * 2, 101, pr)
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
