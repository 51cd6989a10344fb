#!/bin/bash
# round 2, last GPU call: sanity after the oracle gained two stage views (tests only): smoke + a parity subset
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2fin; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 600 python -m pytest tests/test_gpu_backend.py tests/test_gpu_errors.py tests/test_gpu_sharded.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
