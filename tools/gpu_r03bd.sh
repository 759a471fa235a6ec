#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "alternative_launch_paths" 2>&1 | tail -3
