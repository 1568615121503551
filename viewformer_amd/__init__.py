"""viewformer_amd — MI355X-native ViewFormer novel-view hot path.

Scope: VQ-VAE codebook encode -> image-token transformer ->
VQ-VAE decode, behind the reference's model-object protocol.  All device
arithmetic lives in ``csrc/`` (hand-written gfx950 HIP, C-ABI ``libvf_hip.so``);
this package is the Python host side only.
"""
from .config import VQGANConfig, MIGTConfig, load_config  # noqa: F401

__all__ = ['VQGANConfig', 'MIGTConfig', 'load_config']
