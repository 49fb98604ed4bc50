"""neuray_amd: MI355X-native (gfx950) implementation of the NeuRay per-ray render path.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed); the per-ray hot path
is hand-written HIP behind the C ABI of include/neuray_hip.h (libneuray_hip.so, built by
`python -m neuray_amd.build`).  neuray_amd.network mirrors the reference's network.renderer /
network.render_ops call surface.
"""
__version__ = "0.1.0"
