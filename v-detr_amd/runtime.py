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
    DeferredParamGrads.flush()
    DeferredLnGrads.flush()
    DeferredPosEmbedGrads.flush()
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


# ---- a failed stream capture must not leave streams behind in capture state -------------------------------------------------
# torch.cuda.graph ends the capture on its ORIGIN stream when the captured region raises.  Streams the region had forked to (the
# table-gradient side stream, a reducer's side stream, sampling streams) and not yet joined can stay in capture state after that
# (seen with two gloo ranks: the all-reduce inside the region invalidates the capture, and the next pageable host-to-device copy
# of the process fails with "operation not permitted when stream is capturing").  end_stray_captures() asks the HIP runtime for
# each known stream's capture status and ends what is still open.
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
        _hip.hipStreamEndCapture.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
        _hip.hipStreamEndCapture.restype = ctypes.c_int
        _hip.hipGraphDestroy.argtypes = [ctypes.c_void_p]
        _hip.hipGraphDestroy.restype = ctypes.c_int
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


def end_stray_captures(origin, streams=()):
    """After a failed capture on `origin` (a torch.cuda.Stream): bring `origin`, `streams` and this package's own side streams out
    of capture state.  The HIP runtime refuses to end a capture whose forked streams have not joined ("capturing stream has
    unjoined work") and then leaves ALL of them capturing: so every stream that still captures is joined into the origin first
    (an event recorded on it, waited for by the origin), then the origin's capture is ended and its graph dropped.  Returns the
    number of streams that were still capturing."""
    import ctypes
    import torch
    from . import attention as A
    hip = _hip_runtime()
    sides = [st for st in list(streams) + list(getattr(A, "_side_streams", {}).values())
             if isinstance(st, torch.cuda.Stream) and st != origin]
    stray = [st for st in sides if capture_status(st) != 0]
    n = len(stray) + (1 if capture_status(origin) != 0 else 0)
    if capture_status(origin) != 0:
        for st in stray:
            try:
                ev = torch.cuda.Event()
                ev.record(st)
                origin.wait_event(ev)
            except RuntimeError:
                hip.hipGetLastError()
        g = ctypes.c_void_p()
        rc = hip.hipStreamEndCapture(ctypes.c_void_p(origin.cuda_stream), ctypes.byref(g))  # (may return the capture's error: expected)
        if os.environ.get("VDETR_DEBUG_CAPTURE"):
            print(f"[runtime] hipStreamEndCapture(origin) -> {rc}, graph {g.value}", flush=True)
        hip.hipGetLastError()
        if g.value:
            hip.hipGraphDestroy(g)
    for st in stray:  # whatever the origin's end did not take along
        if capture_status(st) != 0:
            g = ctypes.c_void_p()
            hip.hipStreamEndCapture(ctypes.c_void_p(st.cuda_stream), ctypes.byref(g))
            hip.hipGetLastError()
            if g.value:
                hip.hipGraphDestroy(g)
    return n
