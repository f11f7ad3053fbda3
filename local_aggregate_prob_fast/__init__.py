"""Drop-in for the reference package ``local_aggregate_prob_fast``
(model/head/localagg_prob_fast; imported at model/head/gaussian_head.py:32-33)."""
from gaussianformer_amd.local_aggregate import LocalAggregatorProbFast as LocalAggregator  # noqa: F401
from gaussianformer_amd.local_aggregate import _LocalAggregateProb as _LocalAggregate  # noqa: F401
