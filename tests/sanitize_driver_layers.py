"""Layer geometry of torchvision-style ResNets at 224 x 224 in plain Python (no torch): shared by the sanitizer driver and
the host-plan tests."""


def resnet_layers(arch):
    """(Cout, Cin, H, W, k, stride, pad) of every Conv2d / the Linear head of a torchvision-style ResNet at 224 x 224."""
    out = [(64, 3, 224, 224, 7, 2, 3)]
    blocks = {"resnet18": (2, 2, 2, 2), "resnet50": (3, 4, 6, 3), "resnet101": (3, 4, 23, 3)}[arch]
    bottleneck = arch != "resnet18"
    cin, hw = 64, 56
    for stage, nb in enumerate(blocks):
        planes = 64 << stage
        for b in range(nb):
            stride = 2 if (b == 0 and stage > 0) else 1
            if bottleneck:
                out += [(planes, cin, hw, hw, 1, 1, 0), (planes, planes, hw, hw, 3, stride, 1),
                        (planes * 4, planes, hw // stride, hw // stride, 1, 1, 0)]
                if b == 0:
                    out.append((planes * 4, cin, hw, hw, 1, stride, 0))
                cin = planes * 4
            else:
                out += [(planes, cin, hw, hw, 3, stride, 1), (planes, planes, hw // stride, hw // stride, 3, 1, 1)]
                if b == 0 and stage > 0:
                    out.append((planes, cin, hw, hw, 1, stride, 0))
                cin = planes
            hw //= stride
    out.append((1000, cin, 1, 1, 1, 1, 0))
    return out
