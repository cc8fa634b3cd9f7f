#!/bin/bash
# round-6 overlap evidence on ONE box: (1) un-profiled A/B of the two arms (default priorities), (2) the same with the
# dependency chain on a HIGH-priority stream and the weight gradients on a normal one, (3) kernel traces of both arms
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r6/overlap
mkdir -p $O
python3 $R/tools/ab_wgrad_stream.py --steps 10 --rounds 3 --out $O/ab_default.json > $O/ab_default.log 2>&1
python3 $R/tools/ab_wgrad_stream.py --steps 10 --rounds 3 --no-graph --main-priority -1 --priority 0 --out $O/ab_prio.json > $O/ab_prio.log 2>&1
cd /tmp; export TMPDIR=/tmp
for arm in serial side; do
  rm -rf $O/trace_$arm
  rocprofv3 --kernel-trace --output-format csv -d $O/trace_$arm -- python3 $R/tools/ab_wgrad_stream.py --arm $arm --steps 4 > $O/trace_$arm.log 2>&1
  python3 $R/tools/overlap_trace.py $O/trace_$arm $arm > $O/overlap_$arm.json
  rm -rf $O/trace_$arm
done
cd $R
tail -3 $O/ab_default.log | cut -c1-600; tail -3 $O/ab_prio.log | cut -c1-600; cat $O/overlap_serial.json $O/overlap_side.json
