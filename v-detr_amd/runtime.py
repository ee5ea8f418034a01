"""Runtime switches of the MI355X path that are not kernels.

``enable_gemm_tuning``: the projections / FFNs / head MLPs are library GEMMs (hipBLASLt, rocBLAS); which Tensile solution
is fastest for a [1024 x 256] x [256 x 256] fp32 product is not what the libraries' heuristics pick (measured: 15.9 ->
14.8 ms per step with the GEMM shapes of the training step tuned).  PyTorch's TunableOp times the candidate solutions
once per shape during the first (eager, un-captured) steps and caches the winners in a CSV; ``tuning/gfx950_tunableop.csv``
is that cache as measured on an MI355X (TunableOp validates library versions and re-tunes on a mismatch).
"""
import os
import shutil
import tempfile

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
SEED_RESULTS = os.path.join(_HERE, "tuning", "gfx950_tunableop.csv")


def enable_gemm_tuning(rank=0, cache_dir=None):
    """Turn TunableOp on for this process.  Returns the path of the results file in use."""
    import torch.cuda.tunable as tn
    path = os.path.join(cache_dir or tempfile.gettempdir(), f"vdetr_tunableop_rank{rank}.csv")
    if not os.path.exists(path) and os.path.exists(SEED_RESULTS):
        shutil.copy(SEED_RESULTS, path)  # start from the committed measurements
    tn.set_filename(path)
    tn.enable(True)
    tn.tuning_enable(True)
    return path


def defer_weight_grads(enable=True):
    """Compute the weight / bias gradients of the `helpers.linear` layers in shape-batched GEMMs after the backward pass
    instead of one by one inside it (helpers.DeferredParamGrads), and the LayerNorm parameter sums in one launch (add_ln.DeferredLnGrads).  The training loop must call ``flush_weight_grads()``
    after ``loss.backward()`` and before anything reads a ``.grad``."""
    from .add_ln import DeferredLnGrads
    from .helpers import DeferredParamGrads, DeferredPosEmbedGrads
    DeferredParamGrads.enabled = bool(enable)
    if not enable:
        DeferredParamGrads.pending.clear()
        DeferredLnGrads.pending.clear()
        DeferredPosEmbedGrads.pending.clear()


def flush_weight_grads():
    from .add_ln import DeferredLnGrads
    from .helpers import DeferredParamGrads, DeferredPosEmbedGrads
    DeferredParamGrads.flush()
    DeferredLnGrads.flush()
    DeferredPosEmbedGrads.flush()
