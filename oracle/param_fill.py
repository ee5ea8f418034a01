"""Deterministic, construction-order-independent parameter fill shared by ``make_golden.py`` (on the reference's
modules) and the tests (on this repo's modules): a tensor named ``n`` with shape ``s`` always receives the same
values, so fixtures do not have to carry megabytes of weights — and a state-dict name or shape mismatch between
the two implementations shows up as a parity failure.  TEST INFRASTRUCTURE ONLY."""
import zlib

import torch


def _gen(name, salt):
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(name.encode()) + 7919 * salt) & 0x7FFFFFFF)
    return g


@torch.no_grad()
def fill_module(module, salt=0, prefix=""):
    for name, p in module.named_parameters():
        g = _gen(prefix + name, salt)
        if p.dim() >= 2:
            fan_in = p[0].numel()
            p.copy_(torch.randn(p.shape, generator=g) * (0.7 / fan_in ** 0.5))
        elif name.endswith("weight"):  # norm scales
            p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
        else:  # biases
            p.copy_(0.05 * torch.randn(p.shape, generator=g))
    for name, b in module.named_buffers():
        g = _gen(prefix + name, salt)
        if name.endswith("running_mean"):
            b.copy_(0.05 * torch.randn(b.shape, generator=g))
        elif name.endswith("running_var"):
            b.copy_(1.0 + 0.2 * torch.rand(b.shape, generator=g))
        elif name.endswith("gauss_B"):
            b.copy_(torch.randn(b.shape, generator=g))
    return module
