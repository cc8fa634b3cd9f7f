#!/bin/bash
# bf16-mixed step, same box: bf16-stored chained tensors (default) against fp32-stored bf16 values (PARADIS_BF16_STORAGE=0)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r6/amp
mkdir -p $O
for rep in 1 2; do
for st in 1 0; do
  PARADIS_BF16_STORAGE=$st python3 $R/bench.py --amp --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs $AMP_EXTRA > $O/amp_storage${st}_$rep.json 2> $O/amp_storage${st}_$rep.err
  python3 - $O/amp_storage${st}_$rep.json $st <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d.get("roofline", {})
print("storage", sys.argv[2], "ms/step %.2f" % d["ms_per_step"], "samples/s %.1f" % d["value"], "gemm TF %.1f frac %.3f avg launch %.1f us" % (r.get("achieved", 0), r.get("frac", 0), 1e3 * r.get("avg_launch_ms", 0)))
PY
done
done
