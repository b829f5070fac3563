cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash scripts/ab_run.sh scripts/k1_only.py base k1nt base k1nt 2>&1 | grep -v "mode 1" > gpurun_out/e8_k1.txt
cat gpurun_out/e8_k1.txt
for v in base k1nt base k1nt; do
lib=torch_motion_correction_amd/libmcorr.so; [ $v = k1nt ] && lib=variants/k1nt/libmcorr.so
echo "== $v" >> gpurun_out/e8.txt
MCORR_LIB=$PWD/$lib timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], 'warp', r['ms_per_launch'], 'solo', r['ms_per_launch_unshared'], 'whole', r['whole_step_frac'], d['config']['shifts_match_ground_truth'])" >> gpurun_out/e8.txt
done
cat gpurun_out/e8.txt
timeout -k 10 900 python bench.py > gpurun_out/e8_bench_full.json 2> gpurun_out/e8_bench_full.err; echo rc=$?
tail -c 3000 gpurun_out/e8_bench_full.json
