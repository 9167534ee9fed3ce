cd $GRAFT_REPO_ROOT
cp habdec_amd/libhabdec_amd.so /tmp/cur.so
cp habdec_amd/libhabdec_r1.bin habdec_amd/libhabdec_amd.so; echo R1; tools/gpu_kstats.sh r1c5 --workload cfg5 --steps 12 --warmup 3 --sync 2>&1 | head -6; tools/gpu_kstats.sh r1c2 --workload cfg2 --steps 20 --warmup 3 --sync 2>&1 | head -7
cp /tmp/cur.so habdec_amd/libhabdec_amd.so; echo CUR; HD_NO_TAIL=1 tools/gpu_kstats.sh cuc5 --workload cfg5 --steps 12 --warmup 3 --sync 2>&1 | head -6; HD_NO_TAIL=1 tools/gpu_kstats.sh cuc2 --workload cfg2 --steps 20 --warmup 3 --sync 2>&1 | head -7
