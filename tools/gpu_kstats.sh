#!/bin/bash
# Run ON the GPU box: kernel-trace stats of one bench invocation, printing the hd:: kernels.  Usage: tools/gpu_kstats.sh <tag> [bench args...]
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/ks_$tag
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o ks -- python3 $root/bench.py --no-cpu-baseline "$@" > $out/bench.json 2> $out/err.log
rm -f $out/*kernel_trace.csv
python3 - <<P
import csv,glob,re
f=glob.glob('$out/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    m=re.search(r'hd::(?:exact::|(fast)::)?(k_\w+)(<[^>]*>)?',n)
    if m or 'fft' in n: print(f"{(m.group(2)+(m.group(3) or '')+('[fast]' if m.group(1) else '')) if m else n[:40]:40s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}")
P
python3 -c "
import json;d=json.load(open('$out/bench.json'));print('bench', d['value'], 'MS/s', d['ms_per_step'],'ms/step')"
