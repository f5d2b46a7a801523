#!/bin/bash
# repeat the two-rank rehearsal (tools/dist_check.py over gloo, both ranks on cuda:0: two PROCESSES sharing the GPU) N times;
# a frame kernel that is only right when it has the GPU to itself shows up here (looping encoder launches)
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
n=${1:-30}; p=31500; ok=0; bad=0
for rep in $(seq 1 $n); do
  p=$((p+1))
  r=$(timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $p tools/dist_check.py --backend gloo 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ok'])")
  if [ "$r" = "True" ]; then ok=$((ok+1)); else bad=$((bad+1)); echo "run $rep: $r"; fi
done
echo "two-rank rehearsal: ok $ok bad $bad of $n"
