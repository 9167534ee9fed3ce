cd $GRAFT_REPO_ROOT
for v in p3 p0 p0s3; do cp habdec_amd/libs/$v.bin habdec_amd/libhabdec_amd.so; for r in 1 2; do echo "$v: $(python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"])')"; done; done
