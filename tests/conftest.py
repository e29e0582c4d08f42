import json
import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def _usable_cores():
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the GPU box reports every socket core but grants a small CPU share: without this the CPU oracle
    # runs hundreds of threads on a few cores and crawls
    torch.set_num_threads(max(1, min(8, _usable_cores())))


# GPU files in the order a `-x` run must see them: kernels against fp64 / scipy / golden vectors first, then the
# reference-generated pipeline fixtures, the full-size oracle comparisons, and the multi-process REHEARSALS (which compare
# the HIP path with itself) last -- a flake in a rehearsal must never hide an oracle or golden test again (GPUTEST_r04).
GPU_FILE_ORDER = ("test_hip_kernels", "test_hip_pipeline", "test_hip_fullsize", "test_hip_timed_config",
                  "test_hip_long_horizon", "test_hip_extras", "test_hip_odd_layers", "test_hip_determinism", "test_hip_split_bf16",
                  "test_hip_distributed", "test_hip_fullsize_dp")


def _file_rank(item):
    stem = os.path.splitext(os.path.basename(str(item.fspath)))[0]
    return GPU_FILE_ORDER.index(stem) if stem in GPU_FILE_ORDER else -1      # CPU files keep their place at the front


def pytest_collection_modifyitems(config, items):
    items.sort(key=_file_rank)           # stable: the order inside a file is untouched
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


class Tiny:
    """A golden fixture: two tiny ResNets, their spec, data and the reference's outputs."""

    def __init__(self, fname):
        from pleas_merging_amd import resnet as zoo
        from pleas_merging_amd.core.utils import spec_from_json

        self.z = np.load(os.path.join(GOLDEN, fname))
        self.block = str(self.z["block"])
        self.spec = spec_from_json(json.loads(str(self.z["spec_json"])))
        self.m1, self.m2 = (self._model(zoo, p) for p in ("m1", "m2"))

    def _model(self, zoo, prefix):
        m = zoo.tiny_resnet(self.block, (1, 1, 1, 1), num_classes=10, width=4)
        m.load_state_dict(self.state(prefix))
        return m.eval()

    def state(self, prefix):
        return {k[len(prefix) + 1:]: torch.from_numpy(self.z[k]) for k in self.z.files if k.startswith(prefix + "/")}

    def batches(self, prefix="x"):
        n = len([k for k in self.z.files if k.startswith(prefix + "/")])
        return [(torch.from_numpy(self.z["%s/%d" % (prefix, i)]), torch.zeros(4, dtype=torch.long)) for i in range(n)]

    def per_key(self, prefix):
        from pleas_merging_amd.core.utils import Axis

        return {k: torch.from_numpy(self.z["%s/%s" % (prefix, k)]) for k in self.spec}


class LongTrain:
    """tests/golden/tiny_bottleneck_train.npz: weights trained BY THE REFERENCE on the two tiny fixtures for 6 / 21 / 401
    updates (make_golden_bottleneck_train.py).  The training batches are regenerated from their seeds and checked against
    the SHA-256 the generator stored, so the comparison is on the reference's own inputs."""

    def __init__(self):
        self.z = np.load(os.path.join(GOLDEN, "tiny_bottleneck_train.npz"))
        self.n = int(self.z["n_long"])
        self._data = None

    def batches(self, n=None):
        import hashlib

        if self._data is None:
            data = [torch.randn(4, 3, 32, 32, generator=torch.Generator().manual_seed(900 + i)) for i in range(self.n)]
            digest = hashlib.sha256(b"".join(x.numpy().tobytes() for x in data)).hexdigest()
            if digest != str(self.z["xt_sha256"]):
                pytest.skip("torch.randn no longer reproduces the reference run's training batches on this build")
            self._data = [(x, torch.zeros(4, dtype=torch.long)) for x in data]
        return self._data[:n] if n is not None else self._data

    def state(self, prefix):
        return {k[len(prefix) + 1:]: torch.from_numpy(self.z[k]) for k in self.z.files if k.startswith(prefix + "/")}


@pytest.fixture(scope="session")
def long_train():
    return LongTrain()


@pytest.fixture(scope="session")
def tiny_basic():
    return Tiny("tiny_basic.npz")


@pytest.fixture(scope="session")
def tiny_bottleneck():
    return Tiny("tiny_bottleneck.npz")
