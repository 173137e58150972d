cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r03_l_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_l_pytest.log
tail -4 gpurun_out/r03_l_pytest.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r03_l_bench.json 2> gpurun_out/r03_l_bench.err; python -c "
import json; d=json.load(open('gpurun_out/r03_l_bench.json')); print('bench', d['value'], d['ms_per_step'], 'cold', d['cold_value'], d['cold_ms_per_step'], d['roofline']['frac'], d['cpu_baseline'])"
