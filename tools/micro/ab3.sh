cd $GRAFT_REPO_ROOT/r1snap
ks() { tag=$1; shift; out=/tmp/ks_$tag; mkdir -p $out; cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $out -o ks -- python3 $GRAFT_REPO_ROOT/r1snap/bench.py --no-cpu-baseline "$@" > $out/bench.json 2>$out/err.log; python3 - <<P
import csv,glob,re
f=glob.glob('$out/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r['Name']; m=re.search(r'hd::(k_\w+)(<[^>]*>)?',n)
    if m and float(r['AverageNs'])>20000: print(f"  {m.group(1)+(m.group(2) or ''):36s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us")
import json; d=json.load(open('$out/bench.json')); print('  bench', d['value'], d['ms_per_step'])
P
cd $GRAFT_REPO_ROOT/r1snap; }
echo R1 cfg5 sync; ks a --workload cfg5 --steps 12 --warmup 3 --sync
echo R1 cfg2 sync; ks b --workload cfg2 --steps 20 --warmup 3 --sync
echo R1 cfg5 pipe; ks c --workload cfg5 --steps 12 --warmup 3
echo R1 cfg2 pipe; ks d --workload cfg2 --steps 20 --warmup 3
