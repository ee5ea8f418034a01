"""vdetr_amd — MI355X (gfx950) native hot path of V-DETR.

Host side (this package): PyTorch-ROCm modules that keep the reference's module / operator API
(``models/vdetr_transformer.py``, ``models/helpers.py``, ``models/position_embedding.py``,
``third_party/pointnet2/pointnet2_utils.py``, the post-backbone part of ``models/model_vdetr.py``) and its
checkpoint key layout.  Device side: ``lib/libvdetr_hip.so`` (hand-written HIP kernels behind the C-ABI of
``include/vdetr_hip.h``), loaded through ctypes by ``_lib``.  There is no CPU fallback: every op raises when
its tensors are not on the GPU or when the HIP library is missing.
"""
__version__ = "0.1.0"
