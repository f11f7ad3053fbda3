"""Drop-in for ``model/encoder/gaussian_encoder/ops`` of the reference: the import
``from .ops import DeformableAggregationFunction as DAF``
(model/encoder/gaussian_encoder/deformable_module.py:11-14) resolves here."""
from gaussianformer_amd.deformable_aggregation import DeformableAggregationFunction  # noqa: F401
