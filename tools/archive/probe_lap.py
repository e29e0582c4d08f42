import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from scipy.optimize import linear_sum_assignment
from pleas_merging_amd import _lib
if os.environ.get('PLEAS_LIB'):
    _lib.LIB_PATH = os.environ['PLEAS_LIB']; print('lib', _lib.LIB_PATH)
from pleas_merging_amd import hip_ops
rng = np.random.default_rng(0)
def cd(n):
    x = rng.standard_normal((n, 64)).astype(np.float32); y = (x[rng.permutation(n)] + 0.3*rng.standard_normal((n, 64))).astype(np.float32)
    return -np.sqrt(np.maximum(((x[:, None] - y[None])**2).sum(-1), 0)).astype(np.float32)
for n in (256, 512, 1024, 2048):
    for name, a in (("random", rng.standard_normal((n, n)).astype(np.float32)), ("cdist", cd(n))):
        t0 = time.time(); _, want = linear_sum_assignment(a, maximize=True); t_sp = time.time() - t0
        d = torch.from_numpy(a).cuda(); torch.cuda.synchronize()
        hip_ops.solve_lsa_batched([d])
        torch.cuda.synchronize(); t0 = time.time(); out = hip_ops.solve_lsa_batched([d])[0]; torch.cuda.synchronize(); t_hip = time.time() - t0
        print("n=%4d %-6s scipy %.1f ms  hip %.1f ms  equal=%s" % (n, name, t_sp*1e3, t_hip*1e3, bool((out.cpu().numpy() == want).all())))
# rn101-like batch: 7x64, 8x128, 47x256, 7x512, 1x1024, 1x2048 random
sizes = [64]*7 + [128]*8 + [256]*47 + [512]*7 + [1024, 2048]
mats = [torch.from_numpy(rng.standard_normal((n, n)).astype(np.float32)).cuda() for n in sizes]
torch.cuda.synchronize(); t0 = time.time(); hip_ops.solve_lsa_batched(mats); torch.cuda.synchronize()
print("rn101-shaped batch of %d random problems: %.1f ms" % (len(sizes), (time.time() - t0)*1e3))
