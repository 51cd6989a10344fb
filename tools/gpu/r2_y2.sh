#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp LVK_BENCH_BACKEND=gloo LVK_BENCH_ONE_GPU=1
O=gpurun_out/r2y; mkdir -p $O
T="timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511"
( time $T bench.py --gpus 2 --steps 40 --warmup 5 > $O/n2_default.json 2> $O/n2_default.err ) 2>&1 | tail -3; echo "rc $?" >> $O/n2_default.err
python - <<PY
import json
lines=[l for l in open('$O/n2_default.json').read().strip().splitlines() if l.startswith('{')]
print(len(lines), 'json line(s)')
d=json.loads(lines[-1])
print(d['metric'][:60], d['value'], d['n_gpus'], d['scaling'], d['config'].get('parallelism'))
p=d.get('sharded_update_probe')
print('probe:', p.get('value'), p.get('error'), str((p.get('config') or {}).get('shard'))[:300], str(p.get('stderr_tail'))[-300:])
PY
