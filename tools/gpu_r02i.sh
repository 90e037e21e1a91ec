#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_fullsize.py tests/test_gpu_stages.py -m gpu -q -p no:cacheprovider 2>&1 | tail -40
