cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/c5p; timeout 400 rocprofv3 --kernel-trace --stats -f csv -d /tmp/c5p -o k -- python bench.py --voxel-scale 100 --voxels 500000 --steps 20 --warmup 8 --fp32-steps 0 --no-train-entry --no-cpu-baseline --kernel-reps 0 --config5-steps 0 > /dev/null 2>&1
mkdir -p gpurun_out/c5p; cp /tmp/c5p/k_kernel_stats.csv gpurun_out/c5p/kernel_stats.csv
python tools/kstats.py gpurun_out/c5p/kernel_stats.csv 28 40
