cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_host_api.py -x -q -m gpu > gpurun_out/r03_j_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_j_pytest.log
grep -v "^$" gpurun_out/r03_j_pytest.log | head -60
