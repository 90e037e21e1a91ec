"""neo360_amd — MI355X-native implementation of the NeO-360 ray-marching hot path.

Python host code over a C-ABI HIP library (`lib/libneo360_hip.so`, header
`include/neo360_hip.h`).  PyTorch-ROCm supplies device memory, streams and
torch.distributed only; every kernel on the path is hand-written HIP for gfx950.
There is no CPU or eager-PyTorch fallback: calls fail loudly if the library is
missing or the tensors are not on a ROCm device.
"""
__version__ = "0.1.0"
