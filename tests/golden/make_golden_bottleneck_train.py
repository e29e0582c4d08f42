#!/usr/bin/env python3
"""Reference-trained goldens for the Bottleneck fixture, incl. the LONG horizon (build container only):

  PYTHONHASHSEED=0 python tests/golden/make_golden_bottleneck_train.py   ->  tests/golden/tiny_bottleneck_train.npz

tiny_bottleneck.npz holds G3 / G4 / G8 only; this adds, by RUNNING THE REFERENCE on that fixture's models, permutation
and costs: G6 (merged state dicts at ratios 0 / 0.5) and G7 (``train`` after 6, 21 and the drivers' full 401 updates,
pleas_merging.py:367-375, at ratios 0 and 0.5).  The same 401-update run is also stored for the BasicBlock fixture
(tiny_basic.npz has 6 and 21 only).

The 401 training batches are NOT stored (19 MB of incompressible noise): batch i is
``torch.randn(4, 3, 32, 32, generator=manual_seed(900 + i))`` as in make_golden.py, and the fixture carries the SHA-256
of their concatenation so that a test that regenerates them knows it has the reference's inputs.
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (stubs + reference import; generates nothing on import)

N_LONG = 401


def main():
    torch.set_num_threads(4)
    out = {"versions": np.array(json.dumps(mg.VERSIONS)), "n_long": np.int64(N_LONG)}
    data = mg.batches(N_LONG, 4, 900)
    out["xt_sha256"] = np.array(hashlib.sha256(b"".join(x.numpy().tobytes() for x, _ in data)).hexdigest())
    for block, fname, seed in (("bottleneck", "tiny_bottleneck.npz", 20), ("basic", "tiny_basic.npz", 10)):
        z = np.load(os.path.join(HERE, fname))
        m1, m2 = mg.make_pair(block, seed)
        for m, p in ((m1, "m1"), (m2, "m2")):
            m.load_state_dict({k[len(p) + 1:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(p + "/")})
        spec = mg.quiet(mg.ref_spec, m1, ((2, 3, 32, 32),))
        perm = {k: torch.from_numpy(z["am_perm/%s" % k]) for k in spec}
        costs = {k: torch.from_numpy(z["am_cost/%s" % k]) for k in spec}
        for ratio in (0.0, 0.5):
            tag = "%s_r%03d" % (block, int(ratio * 100))
            if block == "bottleneck":
                m3 = mg.quiet(mg.ref_partial_merge, spec, m1, m2, perm, costs, ratio)
                mg.sd_np("merged_%s" % tag, m3.state_dict(), out)
            for max_steps in ((5, 20, N_LONG - 1) if block == "bottleneck" else (N_LONG - 1,)):
                m3 = mg.quiet(mg.ref_partial_merge, spec, m1, m2, perm, costs, ratio)
                m3 = mg.quiet(mg.ref_train, data, m1, m2, m3, spec, perm, costs, ratio, False, max_steps, None,
                              num_classes=10)
                mg.sd_np("trained_%s_s%d" % (tag, max_steps), m3.state_dict(), out)
                print(block, ratio, max_steps, "done")
    path = os.path.join(HERE, "tiny_bottleneck_train.npz")
    np.savez_compressed(path, **out)
    print("tiny_bottleneck_train.npz", os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
