cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for lib in libtpamd.so libtpamd_nocurve.so; do
TPAMD_LIBRARY=$PWD/x-edr-trajectory-planning_amd/csrc/$lib timeout -k 10 400 python tools/pmc_probe.py r03_curve_pmc_$lib "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" -- --no-pipeline --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r03_curve_pmc_$lib.log 2>&1; grep -A9 "^k_sweep" gpurun_out/r03_curve_pmc_$lib.log
done
for so in libtpamd_diag.so libtpamd_diag_nocurve.so; do
DIAG_SO=$so timeout -k 10 300 python tools/gpu_diag.py > gpurun_out/r03_curve_diag_$so.log 2>&1; cut -c1-150 gpurun_out/r03_curve_diag_$so.log
done
