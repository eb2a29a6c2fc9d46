cd $GRAFT_REPO_ROOT
for f in /sys/class/drm/card*/device/numa_node; do echo $f $(cat $f); done
lscpu | grep -i numa
rocm-smi --showtopo 2>/dev/null | tail -15
for mode in pin nopin node0 node1; do
  case $mode in
    pin) pre="";;
    nopin) pre="env DODA_NO_PIN=1";;
    node0) pre="env DODA_NO_PIN=1 taskset -c 0-63,128-191";;
    node1) pre="env DODA_NO_PIN=1 taskset -c 64-127,192-255";;
  esac
  for r in 1 2; do
    $pre python bench.py --no-cpu-baseline --fp32-steps 0 --kernel-reps 1 --steps 150 --warmup 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode', d['ms_per_step'], d['config'].get('host_pinning'))"
  done
done
