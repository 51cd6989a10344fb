#!/bin/bash
# CPU-side sanitizer pass (GPU AddressSanitizer is not available on the pool): (1) the product's HOST code through the host-only checks
# of tests/host/ (they #include backend.hip and exercise its host functions: IMU composition, observation lists, erase counts, the
# moving-start initialiser), (2) the oracle library under the CPU tests that drive it hardest - both with -fsanitize=address,undefined.
# usage: tools/sanitize_cpu.sh     (from the repo root; needs the built objects of larvio_amd/csrc; restores the release oracle afterwards)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"
T=$(mktemp -d /tmp/lvksan.XXXX)
make -C larvio_amd/csrc -j8 -s
OBJS=$(ls larvio_amd/csrc/*.o | grep -v backend.o)
for name in imu_compose_check feature_obs_check erase_count_check init_check; do
  /opt/rocm/bin/hipcc -O1 -g -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Xarch_host -mavx2 -Xarch_host -fsanitize=address,undefined \
      -Xarch_host -fno-omit-frame-pointer -w -I larvio_amd/csrc -c tests/host/$name.hip -o $T/$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -fsanitize=address,undefined $T/$name.o $OBJS -pthread -o $T/$name
  echo "== $name"; ASAN_OPTIONS=detect_leaks=0 $T/$name | tail -2
done
cp oracle/liblvo.so $T/liblvo_release.so
( cd oracle && gcc -O1 -g -march=x86-64-v3 -ffp-contract=off -fno-fast-math -fPIC -std=c11 -w -fopenmp -fsanitize=address,undefined -fno-omit-frame-pointer -shared -o liblvo.so fe_image.c fe_track.c fe_pipeline.c be_*.c -lm )
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=0 OMP_NUM_THREADS=4 \
  python -m pytest tests/test_oracle_frontend.py tests/test_oracle_backend.py tests/test_oracle_decisions.py tests/test_oracle_consistency.py -x -q 2>&1 | grep -a "runtime error\|AddressSanitizer\|passed\|failed" | sort | uniq -c || true
cp $T/liblvo_release.so oracle/liblvo.so; make -C oracle -s -B liblvo.so
rm -rf $T
