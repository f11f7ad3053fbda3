"""Drop-in for the reference package ``local_aggregate_prob`` (model/head/localagg_prob;
imported at model/head/gaussian_head.py:35-36)."""
from gaussianformer_amd.local_aggregate import LocalAggregatorProb as LocalAggregator  # noqa: F401
from gaussianformer_amd.local_aggregate import _LocalAggregateProb as _LocalAggregate  # noqa: F401
