cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
DIAG_SO=libtpamd_reasons.so timeout -k 10 300 python tools/gpu_diag.py > gpurun_out/r03_o_reasons.log 2>&1; grep -E "chain blocks|boundary:|tail: up to|qd/qdd inside|lei\+remaining|fills w/o|blocks from|^B=" gpurun_out/r03_o_reasons.log
