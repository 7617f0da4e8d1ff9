#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_encoder.py -m gpu -q -p no:cacheprovider --timeout 900 -k "fused_and_standalone" 2>&1 | tail -4
