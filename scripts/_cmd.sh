cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 600 python scripts/raw_check.py 2>&1 | tail -9
MCORR_LIB=$PWD/variants/stamp/libmcorr.so REPS=3 timeout -k 10 300 python scripts/raw_time.py 2>&1 | grep stamps | tail -1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/e24_raw -- python3 scripts/raw_time.py > gpurun_out/e24.log 2>&1
f=$(ls -t $(find gpurun_out/e24_raw -name "*kernel_stats.csv") | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:40]:
    if any(k in r['Name'] for k in ('warp_rigid','xc_rows_fwd','raw_stats_k')):
        print(f"{r['Name'][:80]:80s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.1f}")
PY
