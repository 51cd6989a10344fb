#!/bin/bash
# round 2, GPU call J: image stage on its own stream with true overlap (waits for frame f-2), CAQR register T, shard probe
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2j; mkdir -p $O
( time timeout 1200 python -m pytest tests/test_gpu_frontend_sequence.py tests/test_gpu_frontend_edge.py tests/test_gpu_frontend_stages.py tests/test_gpu_vio_driver.py tests/test_gpu_errors.py -m gpu -q -x --durations=5 ) > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
timeout 900 python bench.py --no-cpu-baseline --no-shard-probe > $O/bench_a.json 2> $O/bench_a.err; echo "rc $?" >> $O/bench_a.err
timeout 900 python bench.py --no-cpu-baseline --no-shard-probe > $O/bench_a2.json 2> $O/bench_a2.err
LVK_EKF_TRACE=1 timeout 900 python bench.py --no-cpu-baseline --no-shard-probe --no-device-pass > $O/bench_a_trace.json 2> $O/bench_a_trace.err
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver.json 2> $O/bench_driver.err
timeout 900 python bench.py --sequential --no-cpu-baseline --no-shard-probe --no-device-pass > $O/bench_seq.json 2> $O/bench_seq.err
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_a -- python bench.py --no-cpu-baseline --no-device-pass --no-shard-probe > $O/prof_a.log 2>&1
for db in $(find $O/prof_a -name "*.db" | head -1); do python tools/prof_summary.py $db > $O/a_kernel_stats.csv; python tools/queue_gaps.py $db > $O/a_queue_gaps.txt 2>&1; done
cd /tmp; timeout 600 python $GRAFT_REPO_ROOT/tools/gpu/qr_probe.py > $GRAFT_REPO_ROOT/$O/qr_timing.txt 2>&1; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_backend.py -m gpu -q -k "qr or compress" > $O/pytest_qr.log 2>&1
find $O -name "*.db" -size +20M -delete
tail -3 $O/pytest.log; cut -c1-160 $O/bench_a.json; cut -c1-160 $O/bench_a2.json; cut -c1-160 $O/bench_driver.json; cut -c1-160 $O/bench_seq.json; cat $O/qr_timing.txt; tail -2 $O/pytest_qr.log
