cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r03_m_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_m_pytest.log
tail -4 gpurun_out/r03_m_pytest.log
