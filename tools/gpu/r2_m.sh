#!/bin/bash
# round 2, GPU call M: diagnose the diverging bench run of call L (A/B switches)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2m; mkdir -p $O
B="timeout 600 python bench.py --no-cpu-baseline --no-shard-probe --no-device-pass"
$B > $O/a_default.json 2> $O/a_default.err
LVK_CHOL_FUSED=0 $B > $O/a_nofuse.json 2> $O/a_nofuse.err
LVK_FE_LEGACY_IMAGE_KERNELS=1 $B > $O/a_legacy_img.json 2> $O/a_legacy_img.err
LVK_CHOL_FUSED=0 LVK_FE_LEGACY_IMAGE_KERNELS=1 $B > $O/a_both_off.json 2> $O/a_both_off.err
$B --sequential > $O/a_seq.json 2> $O/a_seq.err
$B > $O/a_default2.json 2> $O/a_default2.err
for f in a_default a_nofuse a_legacy_img a_both_off a_seq a_default2; do echo $f; python - <<PY
import json
try:
    d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1])
    print(d['value'], d['config']['backend'], d['config']['timed_region'], d['roofline']['avg_launch_us'], d['roofline_mfma']['avg_launch_us'])
except Exception as e:
    print('ERR', e); print(open('$O/$f.err').read()[-400:])
PY
done
