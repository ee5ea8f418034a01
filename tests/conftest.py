import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """CPU tests only need the shared object to exist (symbol checks); build it if hipcc is around."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("vdetr_build", os.path.join(ROOT, "v-detr_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if not os.path.exists(mod.LIB):
        mod.build()
    yield


class _OracleExt:
    """pointnet2 `_ext` stand-in backed by the C oracle — FOR TESTS OF THE HOST LOGIC ON CPU ONLY."""

    def __getattr__(self, name):
        import numpy as np
        from oracle import pointnet2_oracle as O

        def call(*args):
            def to_np(a):
                if isinstance(a, torch.Tensor):
                    return a.detach().cpu().numpy()
                if isinstance(a, (list, tuple)) and a and isinstance(a[0], torch.Tensor):
                    return [to_np(x) for x in a]
                return a

            conv = [to_np(a) for a in args]
            if name == "ball_query":
                new_xyz, xyz, radius, nsample = conv
                out = O.ball_query(new_xyz, xyz, radius, nsample)
            elif name == "furthest_point_sampling":
                out = O.furthest_point_sampling(conv[0], conv[1])
            else:
                out = getattr(O, name)(*conv)
            if isinstance(out, list):
                return [torch.from_numpy(np.ascontiguousarray(o)) for o in out]
            if isinstance(out, tuple):
                return tuple(torch.from_numpy(np.ascontiguousarray(o)) for o in out)
            return torch.from_numpy(np.ascontiguousarray(out))

        return call


@pytest.fixture
def cpu_oracle_backend(monkeypatch):
    """Routes the native entry points of the host modules (attention, pointnet2 ops, box decode) to the CPU oracle so that the HOST logic (module wiring,
    box decode, top-k, state-dict layout) can be checked without a GPU.  The product never does this."""
    import vdetr_amd.attention as A
    import vdetr_amd.pointnet2_utils as PU
    from oracle.attention_oracle import fused_attention_reference

    monkeypatch.setattr(A, "fused_attention", fused_attention_reference)
    monkeypatch.setattr(A, "begin_step", lambda device: None)
    monkeypatch.setattr(A, "current_rng", lambda device: None)

    def probs(q, k, **kw):
        return fused_attention_reference(q, k, k, return_probs=True, **kw)[1]

    monkeypatch.setattr(A, "attention_probabilities", probs)
    monkeypatch.setattr(PU, "_ext", _OracleExt())
    import vdetr_amd.box_decode as BD
    from oracle.box_oracle import decode_boxes_reference
    monkeypatch.setattr(BD, "decode_boxes", decode_boxes_reference)
    import vdetr_amd.add_ln as ALN
    from oracle import add_ln_oracle
    monkeypatch.setattr(ALN, "layer_norm", add_ln_oracle.layer_norm)
    monkeypatch.setattr(ALN, "add_dropout_layer_norm", add_ln_oracle.add_dropout_layer_norm)
    yield


def load_golden(name):
    import numpy as np
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


@pytest.fixture(autouse=True)
def _no_leaked_runtime_switches():
    """bench.Trainer turns process-wide switches on (deferred weight gradients); no test may inherit them."""
    yield
    from vdetr_amd.helpers import DeferredParamGrads
    DeferredParamGrads.enabled = False
    DeferredParamGrads.pending.clear()
    from vdetr_amd.add_ln import DeferredLnGrads
    DeferredLnGrads.pending.clear()
    from vdetr_amd.helpers import DeferredPosEmbedGrads
    DeferredPosEmbedGrads.pending.clear()
