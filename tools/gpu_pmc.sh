#!/bin/bash
# Run ON the GPU box: one PMC pass of a bench invocation; prints per-kernel medians of the requested counters for the hd:: kernels.
# Usage: tools/gpu_pmc.sh <tag> "<counters>" [bench args...]
tag=$1; ctr=$2; shift; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc $ctr --output-format csv -d $out -o pmc -- python3 $root/bench.py --no-cpu-baseline --no-also --arith exact "$@" > $out/bench.json 2> $out/err.log
python3 - <<P
import csv,glob,re,collections
f=glob.glob('$out/**/*counter_collection.csv',recursive=True)
if not f: print(open('$out/err.log').read()[-2000:]); raise SystemExit
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    m=re.search(r'hd::(?:exact::|(fast)::)?(k_\w+)(<[^>]*>)?',r['Kernel_Name'])
    if not m: continue
    acc[m.group(2)+(m.group(3) or '')+('[fast]' if m.group(1) else '')][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    n=len(next(iter(v.values())))
    if n < 5: continue
    print(k, {c: round(sorted(x)[len(x)//2]) for c,x in v.items()}, 'launches', n)
P
rm -rf $out/*/*counter_collection.csv $out/*counter_collection.csv 2>/dev/null
