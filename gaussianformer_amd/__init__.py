"""MI355X-native (gfx950 / CDNA4) implementation of GaussianFormer's hot path:
the Gaussian-to-voxel splat (``LocalAggregator``) and the multi-scale deformable image
cross-attention sampler (``DeformableAggregationFunction``).  Hand-written HIP kernels
behind a C-ABI shared library (``include/gf_hip.h``); this package is the Python host side
that mirrors the reference's operator API."""
__version__ = "0.1.0"
