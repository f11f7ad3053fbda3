"""Drop-in for the reference package ``local_aggregate`` (model/head/localagg): the import
``import local_aggregate; local_aggregate.LocalAggregator(**cuda_kwargs)`` at
model/head/gaussian_head.py:38-39 resolves here when this repo root is on ``sys.path``."""
from gaussianformer_amd.local_aggregate import LocalAggregator, _LocalAggregate  # noqa: F401
