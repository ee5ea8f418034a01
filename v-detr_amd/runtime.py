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


def _default_cache_dir():
    """Per user and per library version: a results file written by another torch / ROCm build (or another user's job in
    a shared temp dir) must never be picked up."""
    ver = f"torch{torch.__version__}-hip{torch.version.hip or 'none'}".replace("/", "_")
    root = os.environ.get("VDETR_CACHE_DIR") or os.path.join(os.path.expanduser("~"), ".cache", "vdetr_amd")
    return os.path.join(root, ver)


def enable_gemm_tuning(rank=0, cache_dir=None):
    """Turn TunableOp on for this process.  Returns the path of the results file in use.  The file lives in a private
    (0700) per-user, per-version directory and is unique to this process (rank AND pid: concurrent jobs of one user do not
    share it); it is seeded through a temp file + atomic rename from what earlier runs learned, else from the committed
    MI355X measurements, and rank 0 publishes what this run added when the process exits."""
    import torch.cuda.tunable as tn
    d = cache_dir or _default_cache_dir()
    scratch_dir = None
    try:
        os.makedirs(d, mode=0o700, exist_ok=True)
        if os.path.islink(d) or os.stat(d).st_uid != os.getuid():
            raise OSError("cache directory is a symlink or owned by someone else")
    except OSError:
        d = scratch_dir = tempfile.mkdtemp(prefix="vdetr_tunableop_")  # private by construction; removed at exit
    path = os.path.join(d, f"tunableop_rank{rank}_pid{os.getpid()}.csv")
    # start from what earlier runs of this user / library version measured (shapes outside the committed seed are then
    # tuned once, not on every run and every rank), else from the committed MI355X measurements
    learned = os.path.join(d, "tunableop_learned.csv")
    seed = learned if os.path.exists(learned) and not os.path.islink(learned) else SEED_RESULTS
    if os.path.exists(seed):
        fd, tmp = tempfile.mkstemp(dir=d, prefix=".seed_")
        with os.fdopen(fd, "wb") as dst, open(seed, "rb") as src:
            shutil.copyfileobj(src, dst)
        os.replace(tmp, path)
    tn.set_filename(path)
    tn.enable(True)
    tn.tuning_enable(True)
    # the per-process file is scratch; at exit rank 0 publishes it as the learned file of this cache directory (temp file
    # + atomic rename: concurrent jobs of one user never see a half-written file).  VDETR_TUNABLEOP_SAVE=<file> keeps a
    # copy as well (how tuning/gfx950_tunableop.csv is refreshed on the GPU box).
    if hasattr(tn, "write_file_on_exit"):  # older torch: results written at exit; newer: appended as they are found
        tn.write_file_on_exit(False)
    import atexit

    def _finish(path=path):
        try:
            if hasattr(tn, "write_file"):
                tn.write_file(path)
            keep = os.environ.get("VDETR_TUNABLEOP_SAVE")
            if keep:
                shutil.copy(path, keep)
            if rank == 0 and scratch_dir is None and os.path.getsize(path) > 0:
                fd, tmp = tempfile.mkstemp(dir=d, prefix=".learned_")
                with os.fdopen(fd, "wb") as dst, open(path, "rb") as src:
                    shutil.copyfileobj(src, dst)
                os.replace(tmp, learned)
        except Exception:
            pass
        try:
            os.remove(path)
        except OSError:
            pass
        if scratch_dir is not None:
            shutil.rmtree(scratch_dir, ignore_errors=True)

    atexit.register(_finish)
    _FINISHERS.append(_finish)
    return path


_FINISHERS = []


def finish_gemm_tuning():
    """Publish what this run tuned and remove the per-process scratch NOW.  For callers that leave through ``os._exit`` (which
    skips atexit handlers): bench.py after a distributed run whose captured graphs hold RCCL nodes.  Idempotent."""
    import atexit
    while _FINISHERS:
        fn = _FINISHERS.pop()
        try:
            atexit.unregister(fn)
        except Exception:  # noqa: BLE001
            pass
        fn()


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
        from .attention import DeferredTableGrads
        DeferredTableGrads.pending.clear()
        DeferredTableGrads._begun.clear()
        DeferredTableGrads.buffers.clear()
        DeferredTableGrads._outputs.clear()
        DeferredTableGrads._buffer_keys.clear()
        from . import attention
        attention.SideResults.pending.clear()
        attention.side_late[:] = []  # closures handed over for the end of a backward pass that will not come


def weight_grads_deferred():
    """True while ``defer_weight_grads()`` is on: parameter gradients are then complete only after
    ``flush_weight_grads()`` (dist.GradientReducer launches no bucket from its hooks in that mode)."""
    from .helpers import DeferredParamGrads
    return bool(DeferredParamGrads.enabled)


def flush_weight_grads():
    from .add_ln import DeferredLnGrads
    from .helpers import DeferredParamGrads, DeferredPosEmbedGrads
    from .attention import DeferredTableGrads
    DeferredTableGrads.begin_flush()  # (the table MLPs' backward onto the side stream, behind the table kernels it reads)
    ts_mark("main: flush begins")
    DeferredParamGrads.flush()
    ts_mark("main: flush, own weight gradients done")
    DeferredLnGrads.flush()
    DeferredPosEmbedGrads.flush()
    ts_mark("main: flush, own LayerNorm / position gradients done")
    DeferredTableGrads.flush()  # last: it waits for the side stream's table kernels, which run under the launches above


def flush_weight_grads_phased(phase_of_param, nphases, after_phase):
    """``flush_weight_grads()`` in ``nphases`` steps for a data-parallel step whose gradient buckets are all-reduced one by
    one: a parked weight gradient is computed in the phase of the EARLIEST bucket one of its parameters belongs to
    (``phase_of_param(leaf) -> int``; what cannot be attributed without an autograd walk goes first), so that after phase k
    the buckets 0..k are final and ``after_phase(k)`` may pack and launch bucket k's all-reduce while phase k+1 computes.
    The shape-batched GEMMs split into at most ``nphases`` smaller batches; the values are those of the one-shot flush."""
    from .add_ln import DeferredLnGrads
    from .helpers import DeferredParamGrads, DeferredPosEmbedGrads
    from .attention import DeferredTableGrads
    DeferredTableGrads.begin_flush()  # (onto the side stream; joined before the first bucket leaves)
    DeferredLnGrads.flush()        # a handful of launches whose parameters may sit in any bucket: before the first one leaves
    DeferredPosEmbedGrads.flush()

    def phase(it):
        ph = nphases - 1
        for t in (it[0], it[1]):
            leaves = DeferredParamGrads.leaves_of(t)
            if leaves is None:
                return 0
            for l in leaves:
                ph = min(ph, phase_of_param(l))
        return ph

    for k in range(nphases):
        DeferredParamGrads.flush(select=(lambda it, k=k: phase(it) <= k) if k < nphases - 1 else None)
        if k == 0:  # the table MLPs' parameters may sit in the first bucket: their backward (and the wait for the side stream's
            DeferredTableGrads.flush()  # table kernels, which ran under the GEMMs above) before that bucket leaves
        after_phase(k)


# ---- capture state of a stream ------------------------------------------------------------------------------------------------------
# torch.cuda.graph ends the capture on its origin stream when the captured region raises — and on this HIP runtime that attempt
# fails when a forked stream has not joined ("capturing stream has unjoined work") or the capture was invalidated, after which the
# origin AND the forked streams stay in capture state for good: a second hipStreamEndCapture (from the same thread, through torch
# or directly) is refused with "attempt to terminate a thread-local capture sequence from another thread", and every pageable
# host-to-device copy of the process fails with "operation not permitted when stream is capturing"
# (tools/probes/capture_recovery.py, profiles/r05_capture_recovery.txt).  A process whose capture failed therefore cannot fall back
# in place: bench.py restarts the step in a child process.  capture_status() is what the probe and that decision read.
_hip = None


def _hip_runtime():
    global _hip
    if _hip is None:
        import ctypes
        path = "libamdhip64.so"
        try:  # the copy of the runtime this process already runs on (torch ships its own): a second copy has its own capture state
            with open("/proc/self/maps") as fh:
                for line in fh:
                    if "libamdhip64.so" in line:
                        path = line.split()[-1]
                        break
        except OSError:
            pass
        _hip = ctypes.CDLL(path)
        _hip.hipStreamIsCapturing.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
        _hip.hipStreamIsCapturing.restype = ctypes.c_int
        _hip.hipGetLastError.restype = ctypes.c_int
    return _hip


def capture_status(stream):
    """0 = not capturing, 1 = capturing, 2 = capture invalidated (hipStreamCaptureStatus) of a torch.cuda.Stream"""
    import ctypes
    st = ctypes.c_int(0)
    rc = _hip_runtime().hipStreamIsCapturing(ctypes.c_void_p(stream.cuda_stream), ctypes.byref(st))
    if rc != 0:
        _hip_runtime().hipGetLastError()
        return 2  # (the query itself fails on an invalidated capture)
    return st.value


# ---- timeline marks of a captured step (tools/probes/step_timeline.py) -----------------------------------------------------------
# Off unless VDETR_TS_PROBE=1: `ts_mark(label)` then launches a one-wave kernel on the CURRENT stream that stores the device clock
# into the next slot of a buffer; inside a capture the launches become nodes of the graph and every replay rewrites the slots.
TS_PROBE = os.environ.get("VDETR_TS_PROBE", "0") == "1"
_ts = {"buf": None, "labels": [], "frozen": False}


def ts_mark(label):
    if not TS_PROBE:
        return
    from . import _lib as L
    if _ts["buf"] is None:
        _ts["buf"] = torch.zeros(1024, dtype=torch.int64, device="cuda")
    if _ts["frozen"]:
        return
    i = len(_ts["labels"])
    if i >= 1024:
        return
    _ts["labels"].append(label)
    L.check(L.lib().vdetr_probe_timestamp(_ts["buf"].data_ptr() + 8 * i, L.stream_ptr()), "probe_timestamp")


def ts_reset():
    _ts["labels"], _ts["frozen"] = [], False


def ts_freeze():
    """no further marks (the labels of the captured step stay as they are while it replays)"""
    _ts["frozen"] = True


def ts_read():
    """[(label, microseconds since the first mark)] of the last run / replay"""
    if _ts["buf"] is None:
        return []
    torch.cuda.synchronize()
    v = _ts["buf"][:len(_ts["labels"])].cpu().tolist()
    return [(lab, (t - v[0]) / 100.0) for lab, t in zip(_ts["labels"], v)]
