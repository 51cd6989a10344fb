#!/bin/bash
# round 2, GPU call G: CAQR dense compression (tests + timing), k_dgemm 128-wide K chunks, config 5 / backend-only re-measured
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2h; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_backend.py -m gpu -x -q -s -k "qr_compression or structure_aware or ekf_update or sharded_update" ) > $O/pytest_qr.log 2>&1; echo "rc $?" >> $O/pytest_qr.log
( time timeout 1500 python -m pytest tests -m gpu -q --durations=8 ) > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
python - > $O/qr_timing.txt 2>&1 <<'PY'
import time, numpy as np, ctypes as C, sys
sys.path.insert(0, '.')
import larvio_amd
from larvio_amd import larvio as lv
from larvio_amd._lib import _p
ctx = larvio_amd.Context(0)
L = lv._L()
rng = np.random.default_rng(0)
for rows, cols in ((3536, 442), (18000, 442), (18000, 120), (2000, 442), (900, 442)):
    H = rng.normal(0, 1, (rows, cols)); r = rng.normal(0, 1, rows)
    dH0 = ctx.to_device(H); dr0 = ctx.to_device(r); dH = ctx.alloc(H.nbytes); dr = ctx.alloc(r.nbytes)
    out = C.c_int(0); ts = []
    for rep in range(6):
        L.lvk_memcpy_h2d(ctx.h, C.c_void_p(dH.ptr), _p(H), H.nbytes); L.lvk_memcpy_h2d(ctx.h, C.c_void_p(dr.ptr), _p(r), r.nbytes); ctx.sync()
        t0 = time.perf_counter()
        ctx.check(L.lvk_ekf_compress_qr(ctx.h, C.c_void_p(dH.ptr), cols, rows, cols, C.c_void_p(dr.ptr), C.byref(out))); ctx.sync()
        ts.append(time.perf_counter() - t0)
    fl = 2.0 * rows * cols * cols - 2.0 / 3.0 * cols ** 3
    print("caqr %6d x %4d : %8.1f us  (%.2f TFLOP/s FP64)" % (rows, cols, min(ts) * 1e6, fl / min(ts) / 1e12), flush=True)
PY
timeout 900 python bench.py --no-cpu-baseline --no-shard-probe > $O/bench_a.json 2> $O/bench_a.err; echo "rc $?" >> $O/bench_a.err
timeout 900 python bench.py --backend-only --steps 100 --warmup 6 > $O/bench_be.json 2> $O/bench_be.err; echo "rc $?" >> $O/bench_be.err
timeout 1500 python bench.py --config 5 --steps 200 --warmup 10 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; echo "rc $?" >> $O/bench_c5.err
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof_c5 -- python bench.py --config 5 --steps 100 --warmup 10 --no-cpu-baseline --no-device-pass > $O/prof_c5.log 2>&1
for db in $(find $O/prof_c5 -name "*.db" | head -1); do python tools/prof_summary.py $db > $O/c5_kernel_stats.csv; done
cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_qr -- python $GRAFT_REPO_ROOT/tools/gpu/qr_probe.py > $GRAFT_REPO_ROOT/$O/prof_qr.log 2>&1; cd $GRAFT_REPO_ROOT
for db in $(find $O/prof_qr -name "*.db" | head -1); do python tools/prof_summary.py $db > $O/qr_kernel_stats.csv; done
find $O -name "*.db" -size +20M -delete
tail -5 $O/pytest_qr.log; tail -3 $O/pytest.log; cat $O/qr_timing.txt; cut -c1-160 $O/bench_a.json; cut -c1-200 $O/bench_be.json; cut -c1-160 $O/bench_c5.json
