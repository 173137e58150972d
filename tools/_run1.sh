cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_round3.py -x -q > gpurun_out/r03_a_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_a_pytest.log
tail -5 gpurun_out/r03_a_pytest.log
timeout -k 10 300 python tools/gpu_other_configs.py 4 > gpurun_out/r03_a_cfg4.jsonl 2> gpurun_out/r03_a_cfg4.err; tail -2 gpurun_out/r03_a_cfg4.jsonl
TPAMD_BUCKET=2048 timeout -k 10 300 python tools/gpu_other_configs.py 4 > gpurun_out/r03_a_cfg4_b2048.jsonl 2>> gpurun_out/r03_a_cfg4.err; tail -1 gpurun_out/r03_a_cfg4_b2048.jsonl
TPAMD_BUCKET=0 timeout -k 10 300 python tools/gpu_other_configs.py 4 > gpurun_out/r03_a_cfg4_b0.jsonl 2>> gpurun_out/r03_a_cfg4.err; tail -1 gpurun_out/r03_a_cfg4_b0.jsonl
