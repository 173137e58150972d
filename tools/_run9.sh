cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_multi.py tests/test_gpu_host_api.py -x -q -m gpu > gpurun_out/r03_i_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_i_pytest.log
tail -30 gpurun_out/r03_i_pytest.log
timeout -k 10 600 python tools/profile_other.py r03_cfg4 4 > gpurun_out/r03_cfg4_profile.log 2>&1 && echo cfg4 profiled
