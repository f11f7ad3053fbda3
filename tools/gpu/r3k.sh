#!/bin/bash
timeout 900 python -m pytest tests/test_splat_mfma_gpu.py tests/test_golden_gpu.py tests/test_slab_gpu.py tests/test_ref_parity.py -m gpu -q -x 2>&1 | tail -3
