cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r03_t_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_t_pytest.log
tail -3 gpurun_out/r03_t_pytest.log
timeout -k 10 900 python tools/profile_step.py r03_final > gpurun_out/r03_final_profile.log 2>&1 && echo profiled
cp gpurun_out/r03_final/counters.json profiles/counters.json 2>/dev/null
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r03_final_bench.json 2> gpurun_out/r03_final_bench.err; python -c "
import json; d=json.load(open('gpurun_out/r03_final_bench.json')); print('bench', d['value'], d['ms_per_step'], 'cold', d['cold_value'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline'].get('step_hbm_frac'), d['cpu_baseline']['value'])"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
