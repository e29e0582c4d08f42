#!/usr/bin/env python3
"""How much does the REFERENCE disagree with ITSELF on the degenerate stem?  (build container only)

``conv1`` sees the same input (the image) in both source models and the merged stem is the block
average of the two source stems, so its regression residual is exactly zero in real arithmetic
(reference pleas_merging.py:116-147): what Adam integrates there is the rounding noise of the
convolution kernels.  This script runs the reference's own ``train`` on the tiny_basic fixture under
arithmetic variants that only change summation order -- 1 vs 4 CPU threads, oneDNN convolutions on /
off -- and stores every variant's trained ``conv1.weight`` next to the merged (initial) one in
``stem_spread.npz``.  tests/ use the measured spread (not Adam's maximum travel) to gate the HIP stem.

  PYTHONHASHSEED=0 python tests/golden/make_golden_stem.py
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (stubs + reference import; generates nothing on import)

VARIANTS = (("threads4", 4, True), ("threads1", 1, True), ("threads4_no_onednn", 4, False), ("threads8", 8, True))


def main():
    z = np.load(os.path.join(HERE, "tiny_basic.npz"))
    m1, m2 = mg.make_pair("basic", 10)
    for m, p in ((m1, "m1"), (m2, "m2")):       # the fixture's weights, not the init code's
        m.load_state_dict({k[len(p) + 1:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(p + "/")})
    spec = mg.quiet(mg.ref_spec, m1, ((2, 3, 32, 32),))
    perm = {k: torch.from_numpy(z["am_perm/%s" % k]) for k in spec}
    costs = {k: torch.from_numpy(z["am_cost/%s" % k]) for k in spec}
    n = len([k for k in z.files if k.startswith("xt/")])
    data = [(torch.from_numpy(z["xt/%d" % i]), torch.zeros(4, dtype=torch.long)) for i in range(n)]
    out = {"versions": np.array(json.dumps(mg.VERSIONS)), "variants": np.array([v[0] for v in VARIANTS])}
    for ratio in (0.0, 0.5):
        for steps in (5, 20):
            tag = "r%03d_s%d" % (int(ratio * 100), steps)
            init = mg.quiet(mg.ref_partial_merge, spec, m1, m2, perm, costs, ratio).state_dict()["conv1.weight"]
            out["init_%s" % tag] = init.numpy().copy()
            for name, threads, onednn in VARIANTS:
                torch.set_num_threads(threads)
                torch.backends.mkldnn.enabled = onednn
                m3 = mg.quiet(mg.ref_partial_merge, spec, m1, m2, perm, costs, ratio)
                m3 = mg.quiet(mg.ref_train, data, m1, m2, m3, spec, perm, costs, ratio, False, steps, None, num_classes=10)
                sd = m3.state_dict()
                out["stem_%s/%s" % (tag, name)] = sd["conv1.weight"].numpy().copy()
                # the other tensors must NOT depend on the variant beyond rounding: record the worst rel-fro vs the fixture
                worst = 0.0
                for k, v in sd.items():
                    if k != "conv1.weight" and v.dtype.is_floating_point:
                        w = torch.from_numpy(z["trained_%s/%s" % (tag, k)])
                        worst = max(worst, float((v - w).norm() / (w.norm() + 1e-30)))
                out["others_worst_rel_%s/%s" % (tag, name)] = np.float64(worst)
                d = sd["conv1.weight"] - torch.from_numpy(z["trained_%s/conv1.weight" % tag])
                print(tag, name, "stem max|d| vs fixture %.3e" % float(d.abs().max()), "travel from init %.3e"
                      % float((sd["conv1.weight"] - init).abs().max()), "others worst rel %.2e" % worst)
    torch.backends.mkldnn.enabled = True
    np.savez_compressed(os.path.join(HERE, "stem_spread.npz"), **out)
    print("stem_spread.npz", os.path.getsize(os.path.join(HERE, "stem_spread.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
