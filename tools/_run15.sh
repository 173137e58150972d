cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_round3.py -x -q > gpurun_out/r03_n_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_n_pytest.log
tail -3 gpurun_out/r03_n_pytest.log
timeout -k 10 300 python tools/gpu_diag.py > gpurun_out/r03_n_diag.log 2>&1; grep -E "chain blocks|scalar FindSdd steps|chain steps accepted|whole kernel|guessed|init_carry calls|scalar-step cycles|chain-block cycles|extremal cycles" gpurun_out/r03_n_diag.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03_n_bench.json 2> gpurun_out/r03_n_bench.err; python -c "
import json; d=json.load(open('gpurun_out/r03_n_bench.json')); print('bench', d['value'], d['ms_per_step'], 'cold', d['cold_value'], d['roofline']['kernels'])"
timeout -k 10 300 python tools/gpu_other_configs.py 1 2 3 4 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l)
    if 'case' in d: print(d['case'][:40], d['ms_per_batch'], d['paths_per_s'], d['bit_exact_on_sample'], d.get('kernels_ms'))"
