cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TPAMD_LIBRARY=$PWD/x-edr-trajectory-planning_amd/csrc/libtpamd_k1study.so timeout -k 10 300 python tools/gpu_k1_residency.py > gpurun_out/r03_k1_study.log 2>&1; cat gpurun_out/r03_k1_study.log
