# which switch removes the rare deviating step of 1 cm B4 training runs?  40 runs x 8 steps per variant, in one process each
cd /root/repo
for v in "A=1" "DODA_NO_TILE=1" "DODA_NO_WDMA=1" "PREFETCH=0" "DODA_STATS_TOTALS=0" "DODA_RULEBOOK_GRID=0" "DODA_WGRAD_PAIRS=0" "ORDER=first"; do
echo "== $v"; env $v timeout 900 python tools/traindet.py ${RUNS:-40} 8 2>&1 | grep -E "differs|identical" | tail -4
done
