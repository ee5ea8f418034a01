"""Sparse ResNet backbone with the reference's interface and state-dict keys (`MinkResNet`, models/mink_resnet.py:8-101:
stem `conv1 / norm1`, stages `layer1 .. layer4`), assembled from ``vdetr_amd.minkowski`` instead of MinkowskiEngine."""
from collections import namedtuple

import torch.nn as nn

from . import minkowski as ME

_Arch = namedtuple("_Arch", "block blocks_per_stage")
_ARCHS = {depth: _Arch(block, counts) for depth, block, counts in (
    (18, ME.BasicBlock, (2, 2, 2, 2)), (34, ME.BasicBlock, (3, 4, 6, 3)), (50, ME.Bottleneck, (3, 4, 6, 3)),
    (101, ME.Bottleneck, (3, 4, 23, 3)), (152, ME.Bottleneck, (3, 8, 36, 3)))}


def _stage(block, cin, planes, nblocks):
    """one resolution level: a stride-2 block with a 1x1x1 stride-2 projection on the skip path, then stride-1 blocks"""
    width = planes * block.expansion
    skip = nn.Sequential(ME.MinkowskiConvolution(cin, width, kernel_size=1, stride=2, dimension=3), ME.MinkowskiBatchNorm(width))
    seq = [block(cin, planes, stride=2, downsample=skip, dimension=3)]
    seq += [block(width, planes, stride=1, dimension=3) for _ in range(nblocks - 1)]
    return nn.Sequential(*seq), width


class MinkResNet(nn.Module):
    """MinkResNet(depth, in_channels, inplanes=64, num_stages=4, stem_bn=False); forward(SparseTensor) -> the outputs of the
    `num_stages` levels (tensor strides 4, 8, 16, 32).  The stem halves the resolution (3x3x3, stride 2) and is followed by
    BatchNorm (`stem_bn`) or InstanceNorm, as in the reference."""
    arch_settings = {d: (a.block, a.blocks_per_stage) for d, a in _ARCHS.items()}

    def __init__(self, depth, in_channels, inplanes=64, num_stages=4, stem_bn=False):
        super().__init__()
        if depth not in _ARCHS:
            raise KeyError(f"invalid depth {depth} for resnet")
        if not 1 <= num_stages <= 4:
            raise AssertionError("num_stages must be 1..4")
        arch = _ARCHS[depth]
        self.num_stages, self.inplanes = num_stages, inplanes
        self.conv1 = ME.MinkowskiConvolution(in_channels, inplanes, kernel_size=3, stride=2, dimension=3)
        self.norm1 = (ME.MinkowskiBatchNorm if stem_bn else ME.MinkowskiInstanceNorm)(inplanes)
        self.relu = ME.MinkowskiReLU(inplace=False)
        width = inplanes
        for level, nblocks in enumerate(arch.blocks_per_stage[:num_stages]):
            stage, width = _stage(arch.block, width, inplanes << level, nblocks)
            self.add_module(f"layer{level + 1}", stage)
        self.inplanes = width
        self.init_weights()

    def init_weights(self):
        """Kaiming-normal kernels (fan_out, relu), unit BatchNorm scale, zero shift (mink_resnet.py:53-62)"""
        for mod in self.modules():
            if isinstance(mod, ME.MinkowskiBatchNorm):
                nn.init.ones_(mod.bn.weight)
                nn.init.zeros_(mod.bn.bias)
            elif isinstance(mod, ME.MinkowskiConvolution):
                ME.kaiming_normal_(mod.kernel, mode="fan_out", nonlinearity="relu")

    def forward(self, x):
        x = self.conv1(x)
        fused_stem = isinstance(self.norm1, ME.MinkowskiBatchNorm)
        x = self.norm1(x, act="relu") if fused_stem else self.relu(self.norm1(x))
        feats = []
        for level in range(self.num_stages):
            x = getattr(self, f"layer{level + 1}")(x)
            feats.append(x)
        return feats
