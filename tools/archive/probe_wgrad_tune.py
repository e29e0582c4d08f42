import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pleas_merging_amd import resnet as zoo, hip_ops, _lib
from pleas_merging_amd.core.compiler import get_permutation_spec
from pleas_merging_amd.core.utils import make_identity_perm
from pleas_merging_amd.methods.partial_matching import partial_merge
from pleas_merging_amd.methods.pleas_merging import PleasFitter
dev = torch.device("cuda"); B = 16
torch.manual_seed(0); m1 = zoo.resnet101().to(dev).eval()
torch.manual_seed(1); m2 = zoo.resnet101().to(dev).eval()
spec = get_permutation_spec(m1, ((1, 3, 224, 224),))
perm = make_identity_perm(spec); costs = {k: torch.eye(g.size, device=dev) for k, g in spec.items()}
x = torch.randn(B, 3, 224, 224, device=dev)
for chunks in [int(a) for a in sys.argv[1:]] or [112]:
    _lib.lib().pleas_wgrad_tune(chunks)
    m3 = partial_merge(spec, m1, m2, perm, costs, 0.0)
    fit = PleasFitter(m1, m2, m3, spec, perm, costs, 0.0, 400)
    for _ in range(3): fit.step(x)
    torch.cuda.synchronize(); hip_ops.profile_reset(); hip_ops.profile_enable(True)
    t0 = time.time()
    for _ in range(20): fit.step(x)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 20
    hip_ops.profile_enable(False); p = hip_ops.profile_collect()["conv_wgrad"]
    print("item_chunks=%4d: step %.2f ms, wgrad %.3f ms/launch -> %.1f TF/s" % (chunks, dt * 1e3, p[1] / p[0], p[2] / (p[1] * 1e-3) / 1e12))
    fit.finish()
