#!/bin/bash
# Round-6 final evidence on ONE box: the -m gpu suite (timed), rocprofv3 passes of the four profile tags, bench variants as a
# sanity pass (graph replay, NorMuon, exact GEMMs, checkpointing, bf16-mixed under the graph), then the driver's own bench command.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out/r6 gpurun_out/profiles_out
( time python3 -m pytest tests -q -m gpu --durations=10 ) > gpurun_out/r6/gputest_final7.log 2>&1
tail -4 gpurun_out/r6/gputest_final7.log
bash tools/profile_round.sh r06
bash tools/profile_round.sh r06_cfg3 --workload era5_1.4deg_128x256_S1_B8 --steps 2
bash tools/profile_round.sh r06_cfg4 --workload era5_0.25deg_721x1440_fwd_B1 --steps 2
bash tools/profile_round.sh r06_amp --amp
Q="--steps 5 --warmup 2 --no-cpu-baseline --no-extra-legs --no-other-configs --no-kernel-events"
for v in "--graph" "--optimizer normuon" "--optimizer muon" "--gemm exact" "--checkpoint" "--amp --graph" "--amp --optimizer normuon" "--forward-only"; do
  python3 bench.py $Q $v 2>/dev/null | python3 -c "import json,sys; r=json.loads(sys.stdin.readline()); print('variant $v:', round(r['ms_per_step'],2), 'ms', round(r['value'],1), r['unit'])"
done | tee gpurun_out/r6/bench_variants.txt
python3 bench.py > gpurun_out/r6/bench_final7.json 2> gpurun_out/r6/bench_final7.err
python3 -c "
import json
r=json.loads(open('gpurun_out/r6/bench_final7.json').read().strip().splitlines()[-1])
print(r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['traffic_source']['same_library_build'], r['bf16_mixed_amp']['ms_per_step'])
"
