"""The C-ABI library loads on a CPU-only machine and exports every symbol include/vdetr_hip.h declares."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT


def header_symbols():
    text = open(os.path.join(ROOT, "include", "vdetr_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vdetr_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from vdetr_amd import _lib
    assert header_symbols() == _lib.exported_symbols()


def test_library_exports_every_symbol():
    from vdetr_amd import _lib
    handle = _lib.lib()
    for sym in header_symbols():
        assert hasattr(handle, sym), sym
    assert handle.vdetr_abi_version() == 3


def test_descriptor_layout():
    from vdetr_amd import _lib
    # 6x4 | ptr | 3x4 + pad | 3 ptr | ptr | 2x4 | 2x8 | ptr | 2x4 | ptr | 4x4 | ptr   (ABI 3: the launch-shape fields)
    assert ctypes.sizeof(_lib.AttnDesc) == 168
    assert _lib.AttnDesc.table_grid.offset == 128 and _lib.AttnDesc.kv_halves.offset == 144 and _lib.AttnDesc.kv_img.offset == 152 and _lib.AttnDesc.fwd_sched.offset == 160
    assert _lib.AttnDesc.table.offset == 24 and _lib.AttnDesc.vertices.offset == 48
    assert _lib.AttnDesc.seed.offset == 88 and _lib.AttnDesc.rng_state.offset == 104


def test_argument_errors_do_not_exit():
    """status codes + vdetr_last_error instead of the reference's exit(-1) (cuda_utils.h:32-41)."""
    from vdetr_amd import _lib
    lib = _lib.lib()
    assert lib.vdetr_gather_points_f32(None, None, None, -1, 1, 1, 1, None) == 1
    assert b"negative" in lib.vdetr_last_error()
    assert lib.vdetr_furthest_point_sampling_f32(None, 1, 0, 4, None, None, 0, None) == 1
    assert lib.vdetr_furthest_point_sampling_f32(None, 1, 10, 0, None, None, 0, None) == 0  # m <= 0: no-op
    # the sorted cloud in 64-slot segments of tree-leaf buckets: at most 2 ceil(n / 64) + 64 of them
    assert 40000 * 20 + 256 <= lib.vdetr_fps_workspace_bytes(1, 40000) <= (2 * 625 + 64) * 64 * 20 + 256
    d = _lib.AttnDesc()
    d.kind, d.B, d.H, d.nQ, d.nK = 0, 1, 3, 4, 4
    assert lib.vdetr_attn_fwd_f32(ctypes.byref(d), None, None, None, None, None, None, None, 0, None) == 1
    assert b"4 heads" in lib.vdetr_last_error()


def test_round2_entry_points_reject_bad_arguments():
    """the sparse-convolution / Morton entry points added in round 2: argument errors are status codes with a message"""
    from vdetr_amd import _lib
    lib = _lib.lib()
    assert lib.vdetr_sp_pair_plan_workspace_ints(27, 1000) == 27 * 4
    assert lib.vdetr_sp_pair_plan_i32(None, 0, 10, 10, None, None, None, None, None, None, None) == 1
    assert b"sp_pair_plan" in lib.vdetr_last_error()
    assert lib.vdetr_sp_wgrad_reduce_f32(None, None, 27, 6, None, None) == 1      # elems not a multiple of 4
    assert b"sp_wgrad_reduce" in lib.vdetr_last_error()
    assert lib.vdetr_morton_sort_max() == 16384
    assert lib.vdetr_morton_order_f32(None, 1, 16, None, None, None) == 1
    assert b"morton_order" in lib.vdetr_last_error()
    assert lib.vdetr_sp_pairs_gemm_f32(None, None, None, None, 5, 24, 64, 0, None, None) == 1
    assert b"sp_pairs_gemm" in lib.vdetr_last_error()
    assert lib.vdetr_sp_pairs_gemm_f32(None, None, None, None, 0, 16, 16, 0, None, None) == 0   # no tiles: no-op


def test_round3_attention_backward_entry_points_reject_bad_arguments():
    """vdetr_attn_bwd_kv_f32 / vdetr_attn_bwd_table_f32 / the workgroup-shape setter (round 3): status codes + message."""
    from vdetr_amd import _lib
    lib = _lib.lib()
    d = _lib.AttnDesc()
    d.kind, d.B, d.H, d.nQ, d.nK, d.scale = _lib.VDETR_ATTN_SHARED_KV, 1, 4, 64, 100, 0.125
    # packed operand images: 3 kinds x 4 sub-operands x (hi, lo) x 64 lanes x 16 B per 32-row tile; rows = 4 heads x nQ
    assert lib.vdetr_attn_bwd_kv_workspace_bytes(ctypes.byref(d)) == (64 * 4 // 32) * 3 * 4 * 2 * 64 * 16 + 256
    assert lib.vdetr_attn_bwd_kv_f32(ctypes.byref(d), None, None, None, None, None, None, None, None, None, None, 0, None) == 1
    assert b"attn_bwd_kv" in lib.vdetr_last_error()
    d.kind = _lib.VDETR_ATTN_PER_HEAD
    d.H = 3  # per-head K/V: one problem per (scene, head), rows = queries
    assert lib.vdetr_attn_bwd_kv_workspace_bytes(ctypes.byref(d)) == 3 * (64 // 32) * 3 * 4 * 2 * 64 * 16 + 256
    d.kind, d.H = _lib.VDETR_ATTN_SHARED_KV, 4
    assert lib.vdetr_attn_bwd_table_f32(ctypes.byref(d), None, None, None, 0, None) == 1   # no dS, no table, no bwd_aux
    assert b"attn_bwd_table" in lib.vdetr_last_error()
    assert lib.vdetr_ab_switches() == 0  # the shipped build reads no environment switch


def test_round5_operand_image_entry_points():
    """vdetr_attn_kv_image_bytes / vdetr_attn_pack_kv_f32 and their part-count forms (round 5: the forward's K / V operand images;
    parts 3 = f32 accuracy, 1 = operands rounded to bf16): sizes by the layout (attn_fwd_pipe.hip: a 16-key tile is 6 + 4 pieces
    of 64 lanes x 16 B with three parts, 2 + 2 with one), argument errors as status codes."""
    from vdetr_amd import _lib
    lib = _lib.lib()
    tiles = (4096 + 15) // 16
    assert lib.vdetr_attn_kv_image_bytes(1, 4096) == lib.vdetr_attn_kv_image_parts_bytes(1, 4096, 3) == tiles * 10 * 64 * 16
    assert lib.vdetr_attn_kv_image_parts_bytes(1, 4096, 1) == tiles * 4 * 64 * 16
    assert lib.vdetr_attn_kv_image_parts_bytes(2, 301, 1) == 2 * 19 * 4 * 64 * 16          # ragged key count: whole tiles
    assert lib.vdetr_attn_kv_image_parts_bytes(1, 4096, 2) == 0 and lib.vdetr_attn_kv_image_parts_bytes(0, 4096, 1) == 0
    assert lib.vdetr_attn_pack_kv_f32(None, None, 1, 4096, 64, 64, 1, 0, None, None) == 1
    assert b"attn_pack_kv" in lib.vdetr_last_error()
    assert lib.vdetr_attn_pack_kv_parts_f32(None, None, 1, 4096, 64, 64, 1, 0, 1, None, None) == 1
    assert b"attn_pack_kv" in lib.vdetr_last_error()
    buf = (ctypes.c_char * 64)()
    ptr = ctypes.cast(buf, ctypes.c_void_p)
    ptr = ctypes.c_void_p((ptr.value + 15) & ~15)
    assert lib.vdetr_attn_pack_kv_parts_f32(ptr, ptr, 1, 4096, 64, 64, 1, 0, 2, ptr, None) == 1   # no such part count
    assert b"parts=2" in lib.vdetr_last_error()
    assert lib.vdetr_attn_pack_kv_parts_f32(ptr, ptr, 1, 4096, 62, 64, 1, 0, 1, ptr, None) == 1   # rows shorter than 64 floats
    assert b"attn_pack_kv" in lib.vdetr_last_error()


def test_ops_refuse_cpu_tensors():
    """No CPU fallback: the reference asserts "CPU not supported" (sampling.cpp:36,62,84)."""
    from vdetr_amd import pointnet2_utils as PU
    from vdetr_amd import attention as A
    with pytest.raises(RuntimeError, match="CPU not supported"):
        PU.furthest_point_sample(torch.rand(1, 16, 3), 4)
    with pytest.raises(RuntimeError, match="CPU not supported"):
        PU.gather_operation(torch.rand(1, 4, 16), torch.zeros(1, 2, dtype=torch.int32))
    with pytest.raises(RuntimeError, match="CPU not supported"):
        A.fused_attention(torch.rand(1, 4, 256), torch.rand(1, 8, 64), torch.rand(1, 8, 64), num_heads=4, scale=0.125,
                          shared_kv=True)


def test_build_rejects_the_packed_multiply_form_of_design_4_4b():
    """build.py disassembles the library and refuses `v_pk_mul/fma_f32` with a high-broadcast second source."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("vdetr_build", os.path.join(root, "v-detr_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert b._hi_broadcast(" v[0:1], v[2:3], v[4:5] op_sel:[0,1]")                      # the failing kernel's form
    assert b._hi_broadcast(" v[0:1], v[2:3], v[4:5], v[0:1] op_sel:[1,1,0] op_sel_hi:[0,1,1]")
    assert not b._hi_broadcast(" v[0:1], v[2:3], v[4:5] op_sel:[0,1] op_sel_hi:[1,0]")  # crossed: fine
    assert not b._hi_broadcast(" v[0:1], v[2:3], v[4:5] op_sel_hi:[1,0]")               # low-broadcast: fine
    assert not b._hi_broadcast(" v[0:1], v[2:3], v[4:5] op_sel:[1,0]")                  # first source: covered by parity tests
    assert not b._hi_broadcast(" v[0:1], v[2:3], v[4:5]")
    assert b.check_code_objects(b.build()) >= 15  # every .hip of the library is a code object, none has the form


def test_every_ctypes_mirror_has_the_headers_layout(tmp_path):
    """include/vdetr_hip.h compiled by gcc: sizeof and the offset of every field of every struct that v-detr_amd/_lib.py mirrors
    (the class's docstring names it) — a descriptor that drifts from its mirror is a launch reading garbage, on the GPU only."""
    import re
    import shutil
    import subprocess
    import sys
    from vdetr_amd import _lib
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    src = open(_lib.__file__).read()
    pairs = re.findall(r'class (\w+)\(ctypes\.Structure\):\n\s+"""Mirror of ``(\w+)``', src)
    assert len(pairs) >= 28
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "vdetr_hip.h"', 'int main(void) {']
    want = []
    for cls_name, c_name in pairs:
        cls = getattr(_lib, cls_name)
        lines.append(f'  printf("%zu\\n", sizeof({c_name}));')
        want.append((f"sizeof({c_name})", ctypes.sizeof(cls)))
        for field in cls._fields_:
            name = field[0]
            lines.append(f'  printf("%zu\\n", offsetof({c_name}, {name}));')
            want.append((f"offsetof({c_name}, {name})", getattr(cls, name).offset))
    lines += ['  return 0;', '}']
    c_file = tmp_path / "layout.c"
    c_file.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run([gcc, "-std=c11", "-I", os.path.join(root, "include"), str(c_file), "-o", str(exe)], check=True)
    got = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert len(got) == len(want)
    bad = [(what, int(g), w) for (what, w), g in zip(want, got) if int(g) != w]
    assert not bad, bad
