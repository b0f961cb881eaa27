#!/bin/bash
O=gpurun_out; mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=r04_final
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/${T}_pytest_gpu.log 2>&1; tail -3 $O/${T}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/${T}_smoke.log 2>&1; tail -2 $O/${T}_smoke.log
timeout 400 python bench.py > $O/${T}_bench.log 2>&1; tail -1 $O/${T}_bench.log | cut -c1-600
