cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/gpu_bytes_probe.py 300 2>&1 | tail -9
