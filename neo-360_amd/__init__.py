"""neo360_amd — MI355X-native implementation of the NeO-360 ray-marching hot path.

Python host code over a C-ABI HIP library (`lib/libneo360_hip.so`, header
`include/neo360_hip.h`).  PyTorch-ROCm supplies device memory, streams and
torch.distributed only; every kernel on the inference path is hand-written HIP for gfx950.
There is no CPU or eager-PyTorch fallback: calls fail loudly if the library is
missing or the tensors are not on a ROCm device.

The differentiable training calls (`training.py`) are chains of the library's operators - samplers, encodings, lookups,
MLP / linear-layer GEMMs and compositing, each with a native backward - held together by torch autograd.  NeRF_TP and NeRF run
their MLPs as fused native chains; PixelNeRF and MipNeRF360 compose theirs per layer (`training.linear`) with torch elementwise
ops (ReLU masks, view means, activations) between the GEMMs, and the texel-space projection of the latent uses a library GEMM.
"""
__version__ = "0.1.0"
