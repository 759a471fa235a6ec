#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for r in 1 2; do timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -1; done
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
