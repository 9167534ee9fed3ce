#!/bin/bash
# Run ON the GPU box (through gpurun): kernel-trace stats + two separate PMC passes of the default bench workload.
# Usage: tools/gpu_profile.sh <tag> [bench args...]; outputs under gpurun_out/prof_<tag>/{stats,fetch,write}.
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
args="--steps 20 --warmup 5 --no-cpu-baseline --no-also --arith exact $*"   # (the driver's command; bench.py itself runs ~300 steps around the timed regions)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o stats -- python3 $root/bench.py $args > $out/stats_bench.json 2> $out/stats.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o fetch -- python3 $root/bench.py $args > $out/fetch_bench.json 2> $out/fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o write -- python3 $root/bench.py $args > $out/write_bench.json 2> $out/write.err
# the steady-state summary (launches behind the first 150 of the dominant kernel: the power controller has settled by then) is what the bench line's
# frac_rocprof refers to; rocprofv3's own summary, which includes the ramp, stays beside it
python3 $root/tools/steady_stats.py $(ls $out/stats/*kernel_trace.csv | head -1) $out/stats/steady_kernel_stats.csv 150
rm -f $out/stats/*kernel_trace.csv   # large; the summaries are what gets committed
ls -la $out/*
