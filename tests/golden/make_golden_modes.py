#!/usr/bin/env python3
"""Golden vectors for the reference's other ``merging`` modes (pleas_merging.py:125-144), produced by RUNNING THE
REFERENCE's ``train`` on the tiny_basic fixture (build container only):

  PYTHONHASHSEED=0 python tests/golden/make_golden_modes.py   ->  tests/golden/tiny_modes.npz

reg_mean at ratio 0 (it stacks raw, unpermuted activations: source widths), perm_separatels / perm_mixedls at ratio
0.5 (merged and separate units, gradient mask active); 6 and 21 updates each.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (stubs + reference import; generates nothing on import)


def main():
    torch.set_num_threads(4)
    z = np.load(os.path.join(HERE, "tiny_basic.npz"))
    m1, m2 = mg.make_pair("basic", 10)
    for m, p in ((m1, "m1"), (m2, "m2")):
        m.load_state_dict({k[len(p) + 1:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(p + "/")})
    spec = mg.quiet(mg.ref_spec, m1, ((2, 3, 32, 32),))
    perm = {k: torch.from_numpy(z["am_perm/%s" % k]) for k in spec}
    costs = {k: torch.from_numpy(z["am_cost/%s" % k]) for k in spec}
    n = len([k for k in z.files if k.startswith("xt/")])
    data = [(torch.from_numpy(z["xt/%d" % i]), torch.zeros(4, dtype=torch.long)) for i in range(n)]
    out = {"versions": np.array(json.dumps(mg.VERSIONS))}
    for mode, ratio in (("reg_mean", 0.0), ("perm_separatels", 0.5), ("perm_mixedls", 0.5)):
        for steps in (5, 20):
            m3 = mg.quiet(mg.ref_partial_merge, spec, m1, m2, perm, costs, ratio)
            m3 = mg.quiet(mg.ref_train, data, m1, m2, m3, spec, perm, costs, ratio, False, steps, None, merging=mode, num_classes=10)
            mg.sd_np("trained_%s_r%03d_s%d" % (mode, int(ratio * 100), steps), m3.state_dict(), out)
            print(mode, ratio, steps, "done")
    np.savez_compressed(os.path.join(HERE, "tiny_modes.npz"), **out)
    print("tiny_modes.npz", os.path.getsize(os.path.join(HERE, "tiny_modes.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
