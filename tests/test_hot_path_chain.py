"""All native ops of one training step chained through torch autograd (tools/bench_step.py): feature
pyramid format -> 4 x [SparseConv3D -> deformable_prepare -> DAF] -> fused Gaussian pre-processing -> splat,
backward through everything.  Every leaf (anchors, semantics, instance features, the image pyramid, the
sparse-conv / weights_fc / key-point parameters) must receive a finite, non-zero gradient."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_training_step_chain():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import bench_step
    out = bench_step.run(anchors=3000, steps=1, warmup=1)
    assert out["leaves_without_finite_nonzero_grad"] == [] and out["leaves"] == 23
    assert out["forward_backward_ms"] > out["forward_ms"] > 0
