"""ONE PLeaS update of the HIP path against fp64 autograd of the reference objective on IDENTICAL source activations.

Adam's first updates are sign-like, so trained WEIGHTS compare poorly wherever a gradient is a near-cancellation (the
degenerate stem, rows of fully separate units, deep layers at tiny batches): any two implementations -- the reference on
two conv back ends included -- land +-lr apart there.  The update's own kernels (grouped merge -> fused MFMA forward +
target + residual + loss -> grouped MFMA weight gradient) are therefore held to fp64 directly: the gradient arena after
one ``PleasFitter.step`` vs ``torch.autograd`` in fp64 of ``mean((layer(ip) - op)^2)`` (reference
pleas_merging.py:281-287), with ``ip`` / ``op`` assembled by the oracle from the very taps that update read.
"""
from copy import deepcopy

import torch

from oracle import pleas_oracle as orc


def _rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).norm() / (b.double().cpu().norm() + 1e-30))


def check_update_against_fp64(fit, m3, spec, perm, costs_cpu, ratios, x, num_classes, tol=2e-5, skip=("conv1",)):
    """Applies ``fit.step(x)`` and compares every layer's weight (and bias) gradient and loss with fp64.  Returns
    ``(worst gradient rel-fro, worst fp32-CPU-autograd rel-fro, worst loss rel, largest K)``; afterwards
    ``check_update_against_fp64.levels`` maps every layer to ``loss / mean(target^2)`` in fp64 -- a layer whose input AND
    output groups are (all but) fully separate reproduces its target exactly, like the stem: its loss is ~1e-14 and its
    trained weights are rounding noise integrated by Adam in the reference itself."""
    from pleas_merging_amd.core.utils import get_attr

    kept, release = {}, fit._end_update

    def keep_taps():        # the very tensors this update read (a second source forward may pick other vendor algorithms)
        kept.update(i1=fit.t1_in, i2=fit.t2_in, o1=fit.t1_out, o2=fit.t2_out)
        release()

    fit._end_update = keep_taps
    try:
        fit.step(x)
    finally:
        del fit._end_update
    torch.cuda.synchronize()
    taps = {pl.name: tuple(kept[t][pl.name].detach().double().cpu() for t in ("i1", "i2", "o1", "o2")) for pl in fit.plans}
    kept.clear()
    blocks = orc.spread_blocks(spec, orc.get_blocks(spec, perm, costs_cpu, ratios))
    worst_g = worst_l = worst_cpu = 0.0
    kmax = 0
    check_update_against_fp64.levels = levels = {}      # layer -> loss / mean(target^2): ~0 marks a degenerate layer
    for idx, plan in enumerate(fit.plans):
        ip1, ip2, o1, o2 = taps[plan.name]
        ip, op = orc.layer_targets(lambda _t, o=o1: o, lambda _t, o=o2: o, blocks, plan.name, ip1, ip2, num_classes=num_classes)
        layer = deepcopy(get_attr(m3, plan.name.split("."))).double().cpu()
        for prm in layer.parameters():
            prm.requires_grad_(True)
        loss = ((layer(ip) - op) ** 2).mean()
        grads = torch.autograd.grad(loss, list(layer.parameters()))
        layer32 = deepcopy(get_attr(m3, plan.name.split("."))).float().cpu()       # the same objective in fp32 on the CPU
        for prm in layer32.parameters():
            prm.requires_grad_(True)
        g32 = torch.autograd.grad(((layer32(ip.float()) - op.float()) ** 2).mean(), [layer32.weight])[0]
        gw = plan.gw.permute(0, 3, 1, 2) if plan.kpos else plan.gw
        if plan.name not in skip:        # stem: residual and gradient are rounding noise
            rg, rg_cpu = _rel(gw, grads[0]), _rel(g32, grads[0])
            assert rg < max(tol, 3 * rg_cpu), (plan.name, tuple(gw.shape), rg, rg_cpu)
            worst_g, worst_cpu = max(worst_g, rg), max(worst_cpu, rg_cpu)
            # where the residual all but cancels (rows of fully separate units), fp32 leaves rounding noise of relative
            # size ~1e-7 in `out`: the loss is compared above that floor
            want_l, floor = float(loss.detach()), 1e-10 * float((op ** 2).mean())
            levels[plan.name] = want_l / max(float((op ** 2).mean()), 1e-30)
            rl = abs(float(fit.loss_now[idx]) - want_l) / (want_l + floor)
            assert rl < 1e-5 or abs(float(fit.loss_now[idx]) - want_l) < floor, (plan.name, rl, want_l, floor)
            worst_l = max(worst_l, rl if want_l > 100 * floor else 0.0)
            if plan.gb is not None:
                assert _rel(plan.gb, grads[1]) < max(tol, 3 * rg_cpu), plan.name
        kmax = max(kmax, grads[0][0].numel())
    return worst_g, worst_cpu, worst_l, kmax
