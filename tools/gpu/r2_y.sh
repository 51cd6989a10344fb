#!/bin/bash
# round 2, GPU call Y: rehearsal of the multi-rank control flow on one GPU (two ranks on device 0, gloo): replicas, probe child, sharded paths
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp LVK_BENCH_BACKEND=gloo LVK_BENCH_ONE_GPU=1
O=gpurun_out/r2y; mkdir -p $O
T="timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511"
$T bench.py --gpus 2 --steps 40 --warmup 5 > $O/n2_default.json 2> $O/n2_default.err; echo "rc $?" >> $O/n2_default.err
$T bench.py --gpus 2 --backend-only --sharded --steps 30 --warmup 4 > $O/n2_be_sharded.json 2> $O/n2_be_sharded.err; echo "rc $?" >> $O/n2_be_sharded.err
$T bench.py --gpus 2 --config 5 --sharded --steps 40 --warmup 6 --no-device-pass > $O/n2_c5_sharded.json 2> $O/n2_c5_sharded.err; echo "rc $?" >> $O/n2_c5_sharded.err
for f in n2_default n2_be_sharded n2_c5_sharded; do echo "== $f"; tail -3 $O/$f.err | cut -c1-300; python - <<PY
import json
try:
    lines=[l for l in open('$O/$f.json').read().strip().splitlines() if l.startswith('{')]
    print(len(lines), 'json line(s)')
    d=json.loads(lines[-1])
    print(d['metric'][:60], d['value'], d['n_gpus'], d['scaling'], d['config'].get('parallelism'), str(d['config'].get('shard'))[:200])
    p=d.get('sharded_update_probe')
    if p: print('probe:', p.get('value'), p.get('error'), str((p.get('config') or {}).get('shard'))[:200], str(p.get('stderr_tail'))[-300:])
except Exception as e:
    print('ERR', e)
PY
done
