#!/bin/bash
# Run ON the GPU box (through gpurun): kernel timeline of a workload's batch loop -- start, duration, queue and gap of every hd:: kernel over a slice of
# the sustained region of one bench.py run (tools/micro/timeline.py over a rocprofv3 kernel trace).  Usage: tools/gpu_timeline.sh <workload>  -> gpurun_out/tl_<workload>/timeline.txt
root=$GRAFT_REPO_ROOT; wl=$1
out=$root/gpurun_out/tl_$wl; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out -o tl -- python3 $root/bench.py --workload $wl --steps 40 --warmup 5 --no-cpu-baseline --no-also --arith ${2:-exact} > $out/bench.json 2> $out/err.log
f=$(ls $out/*kernel_trace.csv $out/*/*kernel_trace.csv 2>/dev/null | head -1)
python3 $root/tools/micro/timeline.py $f 120 2>/dev/null | head -90 > $out/timeline.txt
rm -f $f
cat $out/timeline.txt
