cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/c5
for sc in 1 4; do
timeout 600 python bench.py --voxel-scale 100 --voxels 500000 --scenes $sc --steps 20 --warmup 8 --fp32-steps 0 --no-train-entry --no-cpu-baseline --kernel-reps 10 > gpurun_out/c5/bench_s$sc.json 2> gpurun_out/c5/bench_s$sc.err
tail -c 600 gpurun_out/c5/bench_s$sc.err
done
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c5 -o c5 --output-format csv -- python /root/repo/bench.py --voxel-scale 100 --voxels 500000 --scenes 4 --steps 10 --warmup 5 --fp32-steps 0 --no-train-entry --no-cpu-baseline --kernel-reps 2 > /dev/null 2>&1
f=$(find /tmp/prof_c5 -name "*kernel_stats.csv" | head -1); cp $f /root/repo/gpurun_out/c5/kernel_stats_s4.csv
head -30 /root/repo/gpurun_out/c5/kernel_stats_s4.csv | cut -c1-150
