#!/bin/bash
# Round-6 evidence on ONE box: the -m gpu suite (timed), rocprofv3 passes of the default workload, the 128x256 and the
# 0.25-degree legs and the bf16-mixed step, then the driver's own bench command.  Summaries land in gpurun_out/profiles_out.
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/r6 $R/gpurun_out/profiles_out
cd $R
( time python3 -m pytest tests -q -m gpu --durations=15 ) > gpurun_out/r6/gputest_final.log 2>&1
tail -5 gpurun_out/r6/gputest_final.log
bash tools/profile_round.sh r06
bash tools/profile_round.sh r06_cfg3 --workload era5_1.4deg_128x256_S1_B8 --steps 2
bash tools/profile_round.sh r06_cfg4 --workload era5_0.25deg_721x1440_fwd_B1 --steps 2
bash tools/profile_round.sh r06_amp --amp
ls gpurun_out/profiles_out | head -40
# ATen launches per STEADY step: two traces that differ by four timed steps (initialisation fills / copies cancel)
cd /tmp; export TMPDIR=/tmp
for n in 3 7; do
  rm -rf $R/gpurun_out/prof_aten_$n
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_aten_$n -- python3 $R/bench.py --steps $n --warmup 1 --no-cpu-baseline --no-kernel-events --no-extra-legs > $R/gpurun_out/prof_aten_$n.log 2>&1
done
cd $R
python3 - <<'PY' > gpurun_out/profiles_out/r06_steady_step_launches.txt
import csv, glob
def load(n):
    f = glob.glob(f"gpurun_out/prof_aten_{n}/**/*kernel_stats.csv", recursive=True)[0]
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(f))}
a, b = load(3), load(7)
print("launches and kernel time per STEADY training step (32x64 B = 32, default arithmetic): (trace of 7 timed steps - trace of 3) / 4")
tot = totn = aten = atenn = 0
rows = []
for k in b:
    c = (b[k][0] - a.get(k, (0, 0))[0]) / 4.0
    t = (b[k][1] - a.get(k, (0, 0))[1]) / 4.0 / 1e6
    if c <= 0: continue
    rows.append((t, c, k))
    tot += t; totn += c
    if "at::native" in k or "rocclr" in k:
        aten += t; atenn += c
for t, c, k in sorted(rows, reverse=True):
    print("%9.3f ms %7.1f launches  %s" % (t, c, k[:110]))
print("TOTAL %.2f ms in %.0f launches; ATen / runtime copy kernels: %.3f ms in %.1f launches" % (tot, totn, aten, atenn))
PY
tail -3 gpurun_out/profiles_out/r06_steady_step_launches.txt
# SQ counters of the advection kernels on the three grids (velocity sigma ~ what the default model produces at random init)
for cfg in "32x64 32 768 1.0" "128x256 8 768 0.2" "721x1440 1 768 0.3"; do
  tag=$(echo $cfg | cut -d' ' -f1)
  rm -rf gpurun_out/apmc_1 gpurun_out/apmc_2 gpurun_out/apmc_3
  bash tools/advect_pmc.sh $cfg
  python3 tools/pmc_table.py gpurun_out/apmc_1 gpurun_out/apmc_2 gpurun_out/apmc_3 sl_advect > gpurun_out/profiles_out/r06_advect_sq_$tag.txt 2>&1
done
tail -30 gpurun_out/profiles_out/r06_advect_sq_721x1440.txt
