"""vit_amd — MI355X-native (gfx950) hot path of the radar CNN-ViT classifier.

Python host over a C-ABI HIP library (vit_amd/librvit.so, include/rvit.h).  The package has a
single compute path (the HIP kernels); nothing here falls back to eager PyTorch or the CPU.
"""
__version__ = "0.1.0"
