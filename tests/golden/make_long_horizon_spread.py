#!/usr/bin/env python3
"""Yardstick for the long-horizon parity test: the ORACLE AGAINST ITSELF over the drivers' full 401 PLeaS updates.

  python tests/golden/make_long_horizon_spread.py [resnet50]   ->  tests/golden/long_horizon_<arch>_spread.json

Runs ``oracle.train`` twice on the pair of ``tests/long_horizon.py`` (224 x 224, batch 2, full merge): once as it is and
once with oneDNN convolutions switched off -- nothing else changes, i.e. another summation order inside every
convolution.  Records, after updates 1 / 3 / 21 / 101 / 401, the rel-fro distance of every trained tensor between the two
runs and both runs' per-layer losses.  The oracle follows the reference to 1e-5 over 401 updates on the tiny fixtures
(tests/test_oracle_golden.py::test_train_adam_bottleneck_and_long_horizon); this file says what ANY second
implementation of the path can be held to at this depth and horizon.  CPU only, ~8 minutes on 8 cores.
"""
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.dirname(HERE))

from oracle import pleas_oracle as orc  # noqa: E402
import long_horizon as lh  # noqa: E402


def run(pair, perm, costs, onednn):
    m1, m2, spec, _, train = pair
    snaps, losses = {}, {}

    def on_update(idx, layers, per_layer):
        if idx + 1 in lh.SNAPSHOTS:
            snaps[idx + 1] = lh.layer_weights(layers)
            losses[idx + 1] = list(per_layer)

    with torch.backends.mkldnn.flags(enabled=onednn):
        m3 = orc.partial_merge(spec, m1, m2, perm, costs, 0.0)
        t0 = time.time()
        orc.train(train, m1, m2, m3, spec, perm, costs, 0.0, lh.N_UPDATES - 1, on_update=on_update)
        print("oneDNN %s: %d updates in %.0f s" % (onednn, lh.N_UPDATES, time.time() - t0), flush=True)
    return snaps, losses


def main():
    arch = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
    torch.set_num_threads(8)
    pair = lh.build_pair(arch)
    m1, m2, spec, match, _ = pair
    perm, costs = orc.activation_matching(spec, m1, m2, match, 2, accumulate=True)
    a, la = run(pair, perm, costs, True)
    b, lb = run(pair, perm, costs, False)
    out = {"arch": arch, "batch": lh.BATCH, "snapshots": list(lh.SNAPSHOTS), "torch": torch.__version__,
           "what": "rel-fro(oracle with oneDNN off, oracle) per trained tensor after k updates; per-layer losses of both",
           "spread": {}, "loss_default": {}, "loss_variant": {}}
    for k in lh.SNAPSHOTS:
        out["spread"][str(k)] = {name: lh.rel(b[k][name], a[k][name]) for name in a[k]}
        out["loss_default"][str(k)] = la[k]
        out["loss_variant"][str(k)] = lb[k]
        worst = max(out["spread"][str(k)].items(), key=lambda kv: kv[1] if kv[0] != "conv1.weight" else 0.0)
        print("after %3d updates: worst non-stem tensor %s %.2e" % (k, worst[0], worst[1]), flush=True)
    path = os.path.join(HERE, "long_horizon_%s_spread.json" % arch)
    with open(path, "w") as f:
        json.dump(out, f)
    print(path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
