#!/bin/bash
# round 2, GPU call B: structure-aware QR (kernel test, full suite, A/B at config A), first config-5 bench, gather24 calibration
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2b; mkdir -p $O
( time timeout 600 python -m pytest tests/test_gpu_backend.py -m gpu -x -q -s -k "structure_aware" ) > $O/pytest_qr.log 2>&1; echo "rc $?" >> $O/pytest_qr.log
( time timeout 1500 python -m pytest tests -m gpu -q --durations=8 ) > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout 900 python bench.py --no-cpu-baseline > $O/bench_a.json 2> $O/bench_a.err; echo "rc $?" >> $O/bench_a.err
LVK_SPARSE_QR=0 timeout 900 python bench.py --no-cpu-baseline > $O/bench_a_nosparse.json 2> $O/bench_a_nosparse.err
timeout 900 python bench.py --no-cpu-baseline > $O/bench_a2.json 2> $O/bench_a2.err
LVK_EKF_TRACE=1 timeout 900 python bench.py --no-cpu-baseline --no-device-pass > $O/bench_a_trace.json 2> $O/bench_a_trace.err
timeout 1500 python bench.py --config 5 --steps 60 --warmup 10 > $O/bench_c5.json 2> $O/bench_c5.err; echo "rc $?" >> $O/bench_c5.err
LVK_EKF_TRACE=1 timeout 900 python bench.py --config 5 --steps 60 --warmup 10 --no-cpu-baseline --no-device-pass > $O/bench_c5_trace.json 2> $O/bench_c5_trace.err
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_c5 -- python bench.py --config 5 --steps 60 --warmup 10 --no-cpu-baseline --no-device-pass > $O/prof_c5.log 2>&1
for db in $(find $O/prof_c5 -name "*.db" | head -1); do python tools/prof_summary.py $db > $O/c5_kernel_stats.csv; python tools/queue_gaps.py $db > $O/c5_queue_gaps.txt 2>&1; done
timeout 300 tools/lvk_microbench gather24 > $O/microbench_gather.jsonl 2>&1
cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_micro -- $GRAFT_REPO_ROOT/tools/lvk_microbench gather24 > $GRAFT_REPO_ROOT/$O/pmc_micro_g.log 2>&1
cd "$GRAFT_REPO_ROOT"
for c in $(find $O/pmc_micro -name "*counter_collection.csv"); do python tools/pmc_summary.py $c FETCH_SIZE; done > $O/pmc_micro_fetch_size.csv
find $O -name "*.db" -size +20M -delete; find $O -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
tail -3 $O/pytest_qr.log; tail -3 $O/pytest.log; cut -c1-300 $O/bench_a.json; cut -c1-300 $O/bench_c5.json; tail -3 $O/bench_c5.err
