cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_round3.py -x -q > gpurun_out/r03_h_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_h_pytest.log
tail -3 gpurun_out/r03_h_pytest.log
timeout -k 10 300 python tools/gpu_other_configs.py 1 2 4 > gpurun_out/r03_h_other.jsonl 2>>gpurun_out/r03_h.err; cat gpurun_out/r03_h_other.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l)
    if 'case' in d: print(d['case'], d['ms_per_batch'], d['paths_per_s'], d['bit_exact_on_sample'], d.get('kernels_ms'))"
timeout -k 10 600 python tools/profile_other.py r03_cfg4 4 > gpurun_out/r03_cfg4_profile.log 2>&1 && echo cfg4 profiled
timeout -k 10 600 python tools/profile_other.py r03_cfg3 3 > gpurun_out/r03_cfg3_profile.log 2>&1 && echo cfg3 profiled
