cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_round3.py -x -q > gpurun_out/r03_b_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_b_pytest.log
tail -3 gpurun_out/r03_b_pytest.log
TPAMD_BUCKET=0 timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/r03_b_tl -o b0 --output-format csv -- python3 tools/gpu_other_configs.py 4 > gpurun_out/r03_b_tl.log 2>&1
python tools/kernel_timeline.py gpurun_out/r03_b_tl/b0_kernel_trace.csv 14
