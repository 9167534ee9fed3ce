cd $GRAFT_REPO_ROOT
for w in cfg5 cfg2 cfg3; do
  echo "cur $w: $(python bench.py --workload $w --steps 20 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["pipeline"]["launch_path"])')"
  echo "cur $w NO_TAIL: $(HD_NO_TAIL=1 python bench.py --workload $w --steps 20 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["pipeline"]["launch_path"])')"
done
echo "cur cfg4: $(python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["pipeline"]["launch_path"], d["roofline"])')"
