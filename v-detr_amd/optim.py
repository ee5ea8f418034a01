"""Gradient-norm clipping + AdamW on a ``dist.FlatParams`` buffer as one launch (csrc/optim.hip).

Reference: engine.py:105-107 (``clip_grad_norm_(model.parameters(), args.clip_gradient)`` then ``optimizer.step()``) with the
``torch.optim.AdamW`` of optimizer.py:6-26.  Host side of ``vdetr_adamw_clip_f32``: a ``torch.optim.Optimizer`` whose one parameter is
the flat buffer and whose state has torch's AdamW keys (``step``, ``exp_avg``, ``exp_avg_sq``: ``FlatParams.state_dict_per_parameter``
and torch's own ``state_dict`` keep working), so that a script built on the reference's optimizer swaps one constructor.  The sum of
squares behind the norm comes out of the gradient pack's launch (``FlatParams.pack_grads``) where the flat gradient is final there, or
out of one more launch where it was all-reduced afterwards.  GPU only, fp32 only: there is no CPU path.
"""
import ctypes

import torch

from . import _lib as L


class ClipAdamW(torch.optim.Optimizer):
    def __init__(self, flat, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_norm=None, norm_from_pack=True):
        """flat: dist.FlatParams.  max_norm: the ``clip_grad_norm_`` bound (None: no clipping).  norm_from_pack: the flat gradient is
        final when ``flat.pack_grads()`` returns (one rank); False: it changes afterwards (all-reduce) and the norm takes a launch."""
        if not (flat.data.is_cuda and flat.data.dtype == torch.float32):
            raise RuntimeError("ClipAdamW: an fp32 FlatParams buffer on the GPU (there is no CPU path)")
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError(f"ClipAdamW: lr {lr} betas {betas} eps {eps} weight_decay {weight_decay}")
        super().__init__([flat.param], dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self.flat = flat
        self.max_norm = None if max_norm is None else float(max_norm)
        self.norm_from_pack = bool(norm_from_pack)
        dev = flat.data.device
        st = self.state[flat.param]
        st["step"] = torch.zeros((), dtype=torch.float32, device=dev)
        st["exp_avg"] = torch.zeros_like(flat.data)
        st["exp_avg_sq"] = torch.zeros_like(flat.data)
        self._ticket = torch.zeros(4, dtype=torch.int32, device=dev)
        self.grad_norm = torch.zeros((), dtype=torch.float32, device=dev)  # ||g|| of the last step (what clip_grad_norm_ returns)
        self._own_partials = None
        if self.max_norm is not None and self.norm_from_pack:
            flat.want_sumsq = True  # pack_grads() leaves flat.sumsq from now on

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise RuntimeError("ClipAdamW: closures are not supported")
        flat, lib = self.flat, L.lib()
        g = self.param_groups[0]
        st = self.state[flat.param]
        d = L.AdamWDesc()
        d.param, d.grad = flat.data.data_ptr(), flat.grad.data_ptr()
        d.exp_avg, d.exp_avg_sq = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
        d.n = flat.data.numel()
        d.step, d.ticket = st["step"].data_ptr(), self._ticket.data_ptr()
        d.sumsq, d.nsumsq = None, 0
        if self.max_norm is not None:
            part = getattr(flat, "sumsq", None) if self.norm_from_pack else None
            if part is None:  # the flat gradient as it is now
                n = flat.grad.numel()
                nb = lib.vdetr_sumsq_blocks(n)
                if self._own_partials is None:
                    self._own_partials = torch.empty(nb, dtype=torch.float32, device=flat.grad.device)
                part = self._own_partials
                L.check(lib.vdetr_sumsq_f32(flat.grad.data_ptr(), n, part.data_ptr(), nb, L.stream_ptr()), "sumsq")
            d.sumsq, d.nsumsq = part.data_ptr(), part.numel()
            d.max_norm, d.norm_eps = self.max_norm, 1e-6
            d.norm_out = self.grad_norm.data_ptr()
        d.lr, d.beta1, d.beta2 = float(g["lr"]), float(g["betas"][0]), float(g["betas"][1])
        d.eps, d.weight_decay = float(g["eps"]), float(g["weight_decay"])
        L.check(lib.vdetr_adamw_clip_f32(ctypes.byref(d), L.stream_ptr()), "adamw_clip")
        return None
