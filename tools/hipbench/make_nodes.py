"""Writes the tracked-node list of a ResNet-101 matching batch (224x224) in the order the twin graph queues it:
  rn101_nodes.txt          every node contracted:      C HW group
  rn101_nodes_derived.txt  BatchNorm nodes derived:    C HW group source   (source = index of the convolution node, -1 = contracted)
Runs on the CPU (shapes only)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pleas_merging_amd import resnet as zoo
from pleas_merging_amd.core.compiler import get_permutation_spec, trace_with_shapes
am = importlib.import_module("pleas_merging_amd.methods.activation_matching")
HERE = os.path.dirname(os.path.abspath(__file__))
m = zoo.resnet101().eval()
spec = get_permutation_spec(m, ((1, 3, 224, 224),))
keys = list(spec.keys())
node_group = {nax: gi for gi, k in enumerate(keys) for nax in spec[k].node}
gm = trace_with_shapes(m, ((1, 3, 224, 224),))
chains, absorbed = am._bn_chains(gm, m, m)
bn_nodes = {bn.name: bn.args[0].name for bn, _a, _r, _s in chains.values()}
rows, index = [], {}
for node in gm.graph.nodes:
    for nax, gi in node_group.items():
        if nax.key == node.name:
            shape = node.meta["tensor_meta"].shape
            C = shape[nax.axis]; HW = 1
            for d in shape[nax.axis + 1:]: HW *= d
            src = index.get(bn_nodes.get(node.name), -1) if node.name in bn_nodes else -1
            index[node.name] = len(rows)
            rows.append((C, HW, gi, src))
head = "%d %d\n" % (len(rows), len(keys)) + "".join("%d\n" % spec[k].size for k in keys)
open(os.path.join(HERE, "rn101_nodes.txt"), "w").write(head + "".join("%d %d %d\n" % r[:3] for r in rows))
open(os.path.join(HERE, "rn101_nodes_derived.txt"), "w").write(head + "".join("%d %d %d %d\n" % r for r in rows))
print(len(rows), "nodes,", sum(1 for r in rows if r[3] >= 0), "derived")
