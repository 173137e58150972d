cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
C=x-edr-trajectory-planning_amd/csrc
for b in 512 768 896 1024 1280 2048; do echo "== paths $b"; timeout -k 10 300 python tools/gpu_timeline.py r03_s_tl_$b $C/libtpamd.so -- --paths-per-gpu $b 2>&1 | tail -1; done
rm -rf gpurun_out/r03_s_tl*
