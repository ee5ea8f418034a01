#!/usr/bin/env python3
"""Would rocBLAS' pointer-array batched GEMM spare the step its torch.stack copies (64 x [1024, 256] operands = 2 x 64 MB per step)?
dW[i] = G_i^T X_i for 64 separately allocated (G_i, X_i): rocblas_sgemm_batched through ctypes on torch's own librocblas, against
torch.stack + torch.bmm; device times from captured graphs (also: does the call survive a stream capture?)."""
import ctypes, os, sys
import torch

lib_path = os.path.join(os.path.dirname(torch.__file__), "lib", "librocblas.so")
rb = ctypes.CDLL(lib_path)
h = ctypes.c_void_p()
assert rb.rocblas_create_handle(ctypes.byref(h)) == 0
N_, T_ = 111, 112  # rocblas_operation_none / transpose

n, rows, C = 64, 1024, 256
torch.manual_seed(0)
Gs = [torch.randn(rows, C, device="cuda") for _ in range(n)]
pad = [torch.empty(12345, device="cuda") for _ in range(n)]  # (keep the operands apart)
Xs = [torch.randn(rows, C, device="cuda") for _ in range(n)]
out = torch.empty(n, C, C, device="cuda")
pa = torch.tensor([x.data_ptr() for x in Xs], dtype=torch.int64).cuda()
pb = torch.tensor([g.data_ptr() for g in Gs], dtype=torch.int64).cuda()
pc = torch.tensor([out[i].data_ptr() for i in range(n)], dtype=torch.int64).cuda()
alpha, beta = ctypes.c_float(1.0), ctypes.c_float(0.0)
rb.rocblas_sgemm_batched.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                     ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]

def own():
    st = torch.cuda.current_stream().cuda_stream
    assert rb.rocblas_set_stream(h, ctypes.c_void_p(st)) == 0
    # row-major dW [out, in] = G^T X  ==  column-major [in x out] = X_cm [in x rows] * (G_cm [out x rows])^T
    rc = rb.rocblas_sgemm_batched(h, N_, T_, C, C, rows, ctypes.byref(alpha), ctypes.c_void_p(pa.data_ptr()), C, ctypes.c_void_p(pb.data_ptr()), C,
                                  ctypes.byref(beta), ctypes.c_void_p(pc.data_ptr()), C, n)
    assert rc == 0, rc
    return out

def ref():
    return torch.bmm(torch.stack(Gs).transpose(1, 2), torch.stack(Xs))

a = own().clone(); b = ref()
print("max |diff| / max |ref|:", float((a - b).abs().max() / b.abs().max()))

def t(fn, reps=10):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps): fn()
    torch.cuda.current_stream().wait_stream(s)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * reps) * 1000

print(f"rocblas_sgemm_batched (pointer arrays): {t(own):7.1f} us")
print(f"torch.stack x 2 + torch.bmm:            {t(ref):7.1f} us")
stk = (torch.stack(Gs), torch.stack(Xs))
print(f"torch.bmm on stacked operands alone:    {t(lambda: torch.bmm(stk[0].transpose(1, 2), stk[1])):7.1f} us")
