cd $GRAFT_REPO_ROOT
timeout -k 10 900 python bench.py --steps 40 > gpurun_out/e23_bench.json 2> gpurun_out/e23_bench.err; echo rc=$?
python - <<'PY'
import json
d=json.loads(open('gpurun_out/e23_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['whole_step_frac'], d['roofline']['frac'], d['cpu_baseline'])
print({k: d['raw_u8'][k] for k in ('ms_per_step','via_fp32_movie_ms_per_step','shifts_match_ground_truth')})
print({k: d['fp16_storage'][k] for k in ('ms_per_step','shifts_match_ground_truth')})
PY
