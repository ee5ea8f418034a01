"""TEST INFRASTRUCTURE — torch restatement of residual add + dropout + LayerNorm (reference
models/vdetr_transformer.py:531-568: ``tgt = tgt + self.dropoutN(tgt2); tgt2 = self.norm(tgt)``) with the signatures of
v-detr_amd/add_ln.py.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline import this module.
``keep`` lets a test inject the kernel's own dropout keep-mask (the generators differ, the arithmetic must not)."""
import torch.nn.functional as F


def layer_norm(x, ln, ln2=None):
    out = F.layer_norm(x, ln.normalized_shape, ln.weight, ln.bias, ln.eps)
    if ln2 is None:
        return out
    return out, F.layer_norm(x, ln2.normalized_shape, ln2.weight, ln2.bias, ln2.eps)


def add_dropout_layer_norm(x, r, drop, ln, ln2=None, salt=0, also_drop=None, keep=None):
    if keep is not None:
        p2 = also_drop.p if (also_drop is not None and also_drop.training) else 0.0
        y = x + r * keep / ((1.0 - drop.p) * (1.0 - p2))
    else:
        if also_drop is not None:  # the reference applies the two dropouts one after the other
            r = also_drop(r)
        y = x + (drop(r) if drop is not None else r)
    res = layer_norm(y, ln, ln2)
    return (y, res) if ln2 is None else (y,) + tuple(res)
