# Everything the round-end numbers come from, in one gpurun call (tests, rocprofv3 counter passes, bench lines,
# other configurations, smoke):  gpurun --timeout 1200 -- "bash tools/measure_final.sh"
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r03_l_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_l_pytest.log
tail -3 gpurun_out/r03_l_pytest.log
grep -q " passed" gpurun_out/r03_l_pytest.log || exit 1
timeout -k 10 900 python tools/profile_step.py r03_final7 > gpurun_out/r03_final7_profile.log 2>&1 && echo profiled
cp gpurun_out/r03_final7/counters.json profiles/counters.json 2>/dev/null
find gpurun_out -name "*.db" -delete
for rep in 1 2 3; do
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r03_final8_bench_$rep.json 2> gpurun_out/r03_final8_bench.err; python -c "
import json; d=json.load(open('gpurun_out/r03_final8_bench_$rep.json')); print('bench', d['value'], d['ms_per_step'], 'cold', d['cold_value'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['roofline'].get('step_hbm_frac'), d['roofline']['traffic'], d['cpu_baseline']['value'])"
done
timeout -k 10 300 python bench.py --steps 300 --warmup 5 --no-cpu-baseline > gpurun_out/r03_final8_bench_300.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r03_final8_bench_300.json')); print('bench300', d['value'], d['ms_per_step'])"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pipeline > gpurun_out/r03_final8_bench_unpiped.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r03_final8_bench_unpiped.json')); print('unpiped20', d['value'], d['ms_per_step'])"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --pipeline 2 > gpurun_out/r03_final8_bench_mode2.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r03_final8_bench_mode2.json')); print('mode2', d['value'], d['ms_per_step'])"
timeout -k 10 400 python tools/gpu_other_configs.py 1 2 3 4 > gpurun_out/r03_final8_other.jsonl 2>>gpurun_out/r03_l.err; cat gpurun_out/r03_final8_other.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l)
    if 'case' in d: print(d['case'], d['ms_per_batch'], d['paths_per_s'], d['bit_exact_on_sample'])"
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout -k 10 400 python tools/gpu_fuzz.py 200 341 > gpurun_out/r03_fuzz_341.json 2> gpurun_out/r03_fuzz_341.err; cut -c1-200 gpurun_out/r03_fuzz_341.json
timeout -k 10 400 python tools/gpu_fuzz.py 200 342 > gpurun_out/r03_fuzz_342.json 2> gpurun_out/r03_fuzz_342.err; cut -c1-200 gpurun_out/r03_fuzz_342.json
