#!/bin/bash
# Run ON the GPU box (through gpurun): kernel-trace stats + two separate PMC passes of the default bench workload.
# Usage: tools/gpu_profile.sh <tag> [bench args...]; outputs under gpurun_out/prof_<tag>/{stats,fetch,write}.
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
args="--steps 60 --warmup 10 --no-cpu-baseline --no-also $*"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o stats -- python3 $root/bench.py $args > $out/stats_bench.json 2> $out/stats.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o fetch -- python3 $root/bench.py $args > $out/fetch_bench.json 2> $out/fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o write -- python3 $root/bench.py $args > $out/write_bench.json 2> $out/write.err
rm -f $out/stats/*kernel_trace.csv   # large; the stats summary is what gets committed
ls -la $out/*
