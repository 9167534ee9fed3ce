#!/bin/bash
# Run ON the GPU box: the default command N times in a row (one process each), sustained / cold / kernel of both modes per run -- is the fast leg's sustained
# region (behind the exact leg's all-stream self-check in the same interpreter) as steady as its kernel?  Usage: tools/micro/r06_default_repeat.sh [N]
cd $GRAFT_REPO_ROOT
o=gpurun_out/default_repeat; mkdir -p $o
for i in $(seq 1 ${1:-3}); do
  timeout 900 python3 bench.py > $o/line_$i.json 2> $o/err_$i.txt
  python3 - $o/line_$i.json <<'P'
import json, sys
d = json.load(open(sys.argv[1])); f = d["fast"]
print("exact %.4f ms (cold %.4f, kernel %.4f) host %s | fast %.4f ms (cold %.4f, kernel %.4f) host %s" % (d["ms_per_step"], d["cold"]["ms_per_step"], d["roofline"]["avg_launch_ms"],
      d["pipeline"]["host_us_per_step"], f["ms_per_step"], f["cold"]["ms_per_step"], f["roofline"]["avg_launch_ms"], f.get("host_us_per_step")))
P
done
