# A/B: round-1 library vs current on the secondary workloads (run on the GPU box)
cd $GRAFT_REPO_ROOT
cp habdec_amd/libhabdec_amd.so /tmp/cur.so
for w in cfg5 cfg2 cfg3; do
  cp habdec_amd/libhabdec_r1.bin habdec_amd/libhabdec_amd.so; echo "r1  $w: $(python bench.py --workload $w --steps 20 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
  cp /tmp/cur.so habdec_amd/libhabdec_amd.so; echo "cur $w: $(python bench.py --workload $w --steps 20 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
  echo "cur $w NO_TAIL: $(HD_NO_TAIL=1 python bench.py --workload $w --steps 20 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
done
