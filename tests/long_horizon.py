"""Shared set-up of the LONG-HORIZON parity run: a ResNet pair at 224 x 224, batch 2, the drivers' full 401 PLeaS updates
(reference pleas_merging.py:367-375), snapshots after updates 1 / 3 / 21 / 101 / 401.

Used by ``tests/golden/make_long_horizon_spread.py`` (CPU: the oracle against itself with oneDNN convolutions off -- the
yardstick) and by ``tests/test_hip_long_horizon.py`` (MI355X: HIP path against the oracle)."""
import torch

SNAPSHOTS = (1, 3, 21, 101, 401)        # after this many updates
N_UPDATES = 401
BATCH = 2


def build_pair(arch: str = "resnet50", n_updates: int = N_UPDATES, batch: int = BATCH):
    """(m1, m2, spec, matching batches, training batches): CPU, seeded, BatchNorm calibrated on the matching batches."""
    from pleas_merging_amd import resnet as zoo
    from pleas_merging_amd.core.compiler import get_permutation_spec

    g = torch.Generator().manual_seed(77)
    match = [(torch.randn(batch, 3, 224, 224, generator=g), torch.zeros(batch)) for _ in range(3)]
    train = [(torch.randn(batch, 3, 224, 224, generator=g), torch.zeros(batch)) for _ in range(n_updates)]
    models = []
    for seed in (0, 1):
        torch.manual_seed(seed)
        m = zoo.MODELS[arch](num_classes=1000)
        zoo.calibrate_bn(m, [d[0] for d in match])
        models.append(m.eval())
    spec = get_permutation_spec(models[0], ((1, 3, 224, 224),))
    return models[0], models[1], spec, match, train


def layer_weights(layers) -> dict:
    """``{name.weight / name.bias: tensor}`` of the oracle's layer copies at a snapshot."""
    out = {}
    for name, layer in layers.items():
        for k, v in layer.state_dict().items():
            out["%s.%s" % (name, k)] = v.detach().clone()
    return out


def rel(a, b) -> float:
    return float((a.double().cpu() - b.double().cpu()).norm() / (b.double().cpu().norm() + 1e-30))
