"""hawkeye_amd - MI355X-native high-order pooling / attention-pooling heads for
Hawkeye (BCNN, CBCNN, Fast MPN-COV, AP-CNN, OSME) behind Hawkeye's own
`model.registry` plugin surface.  Hand-written HIP kernels for gfx950 in
`hawkeye_amd/csrc` (C ABI: include/hawkeye_hip.h); PyTorch-ROCm is used for
device memory, streams, the convolutional backbones and torch.distributed."""
__version__ = '0.1.0'
