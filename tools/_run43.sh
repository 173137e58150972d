cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r03_j_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_j_pytest.log
tail -3 gpurun_out/r03_j_pytest.log
grep -q " passed" gpurun_out/r03_j_pytest.log || exit 1
timeout -k 10 900 python tools/profile_step.py r03_final2 > gpurun_out/r03_final2_profile.log 2>&1 && echo profiled
cp gpurun_out/r03_final2/counters.json profiles/counters.json 2>/dev/null
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r03_final2_bench.json 2> gpurun_out/r03_final2_bench.err; python -c "
import json; d=json.load(open('gpurun_out/r03_final2_bench.json')); print('bench', d['value'], d['ms_per_step'], 'cold', d['cold_value'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline'].get('step_hbm_frac'), d['cpu_baseline']['value'])"
timeout -k 10 300 python bench.py --steps 300 --warmup 5 --no-cpu-baseline > gpurun_out/r03_final2_bench_300.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r03_final2_bench_300.json')); print('bench300', d['value'], d['ms_per_step'])"
timeout -k 10 300 python bench.py --steps 300 --warmup 5 --no-cpu-baseline --no-pipeline > gpurun_out/r03_final2_bench_unpiped.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r03_final2_bench_unpiped.json')); print('unpiped300', d['value'], d['ms_per_step'])"
timeout -k 10 400 python tools/gpu_other_configs.py 1 2 3 4 > gpurun_out/r03_final2_other.jsonl 2>>gpurun_out/r03_j.err; cat gpurun_out/r03_final2_other.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l)
    if 'case' in d: print(d['case'], d['ms_per_batch'], d['paths_per_s'], d['bit_exact_on_sample'])"
timeout -k 10 600 python tools/profile_other.py r03b_cfg4 4 > gpurun_out/r03b_cfg4_profile.log 2>&1 && echo cfg4 profiled
timeout -k 10 600 python tools/profile_other.py r03b_cfg3 3 > gpurun_out/r03b_cfg3_profile.log 2>&1 && echo cfg3 profiled
find gpurun_out -name "*.db" -delete
timeout -k 10 600 python tools/gpu_plan_bench.py 1024 7 16 1 3 > gpurun_out/r03b_plan_bench.json 2> gpurun_out/r03b_plan_bench.err; cat gpurun_out/r03b_plan_bench.json | cut -c1-600
timeout -k 10 600 python tools/gpu_plan_bench.py 1024 7 16 0 12 > gpurun_out/r03b_plan_bench_w12.json 2>> gpurun_out/r03b_plan_bench.err; cat gpurun_out/r03b_plan_bench_w12.json | cut -c1-400
DIAG_SO=libtpamd_diag.so DIAG_HIST=gpurun_out/r03b_hist.json timeout -k 10 300 python tools/gpu_diag.py > gpurun_out/r03b_diag.log 2>&1; tail -1 gpurun_out/r03b_diag.log
