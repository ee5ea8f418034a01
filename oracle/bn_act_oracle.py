"""TEST INFRASTRUCTURE — torch restatement of BatchNorm1d + ReLU + Dropout (reference models/helpers.py:74-141, the hidden
blocks of GenericMLP) with the signature of v-detr_amd/bn_act.py:bn_act.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline import this module.  ``keep`` injects the kernel's own dropout keep-mask."""
import torch.nn.functional as F


def bn_act(x, weight, bias, running_mean, running_var, training, eps, momentum, relu=True, dropout_p=0.0, salt=0,
           pre_bias=None, counters=(), keep=None):
    if pre_bias is not None:  # the reference adds the convolution bias before the BatchNorm
        x = x + pre_bias.view(1, -1, 1)
    for c in counters:
        if training:
            c.add_(1)
    y = F.batch_norm(x, running_mean, running_var, weight, bias, training, momentum, eps)
    if relu:
        y = F.relu(y)
    if training and dropout_p > 0.0:
        y = y * keep / (1.0 - dropout_p) if keep is not None else F.dropout(y, dropout_p, True)
    return y


def relu_dropout(x, drop, salt=0, keep=None):
    y = F.relu(x)
    if keep is not None:
        return y * keep / (1.0 - drop.p)
    return drop(y) if drop is not None else y
