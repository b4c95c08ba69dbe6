cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_proj; rm -rf $OUT; mkdir -p $OUT; cd $R
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --workload proj1080 --batch 64 --steps 20 --warmup 5 --streams 1 --no-cpu-baseline --no-other-configs --no-boundary > $OUT/log.txt 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1); cut -d, -f1-4 $f | cut -c1-110 | head -16
