#!/bin/bash
# Multi-GPU scaling run of the benchmark on ONE node (run from the repo root on a box with N MI355X):
#   tools/run_scale.sh [max_gpus=8] [steps=10]
# For N in 1 2 4 8 (<= max_gpus): one process per GPU through torch.distributed.run (the launcher starts
# before anything touches the GPU), RCCL over xGMI, weak scaling (32 samples per GPU).  Prints the JSON
# line of every run, the algorithm/protocol RCCL chose for the 240 MB gradient all-reduce
# (NCCL_DEBUG=INFO), and an A/B of the DDP bucket size and static_graph at the largest N.
# No 8-GPU node was available to the builder in rounds 1-2: the 1 -> 8 curve is measured by the driver.
MAXN=${1:-8}; STEPS=${2:-10}
export HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out/scale; mkdir -p $OUT
run() {  # n, extra bench args, tag
  local n=$1 extra=$2 tag=$3 port=$((29500 + RANDOM % 200))
  if [ "$n" = 1 ]; then
    python3 bench.py --gpus 1 --steps $STEPS --warmup 3 --no-cpu-baseline --no-exact-leg $extra > $OUT/$tag.json 2> $OUT/$tag.err
  else
    NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,COLL python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n \
      --master-addr 127.0.0.1 --master-port $port bench.py --gpus $n --steps $STEPS --warmup 3 \
      --no-cpu-baseline --no-exact-leg $extra > $OUT/$tag.json 2> $OUT/$tag.err
  fi
  echo "== $tag"; tail -n 1 $OUT/$tag.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(' samples/s %.1f  ms/step %.1f  n_gpus %d  ddp %s' % (d['value'], d['ms_per_step'], d['n_gpus'], d['config'].get('ddp')))"
  grep -m 3 -E "Algo|algorithm|Channel|Ring|Tree" $OUT/$tag.err | sed 's/^/   rccl: /'
}
NS=""; for n in 1 2 4 8; do [ $n -le $MAXN ] && NS="$NS $n"; done
for n in $NS; do run $n "" n$n; done
python3 - <<PY
import json
vals = {}
for n in "$NS".split():
    try: vals[int(n)] = json.loads(open("$OUT/n%s.json" % n).read().strip().splitlines()[-1])["value"]
    except Exception as e: print("n", n, "failed:", e)
for n, v in sorted(vals.items()):
    print("N=%d  %.1f samples/s  efficiency %.3f" % (n, v, v / (n * vals[1]) if 1 in vals else float("nan")))
PY
N=$(echo $NS | awk '{print $NF}')
if [ "$N" -gt 1 ]; then
  for mb in 8 32 128; do run $N "--bucket-mb $mb" n${N}_bucket$mb; done
  run $N "--bucket-mb 32 --static-graph" n${N}_static
fi
