#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
ERASOR_HIP_SORT_STAMPS=1 timeout 200 python bench.py --no-cpu-baseline --no-extra-workloads --steps 20 --warmup 5 2>&1 >/dev/null | grep -i "slowest\|stamp" | tail -12
