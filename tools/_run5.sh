cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
run() { tag="$1"; shift; env "$@" timeout -k 10 300 python tools/gpu_other_configs.py 4 2>>gpurun_out/r03_e.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['buckets'], d['ms_per_batch'], d['paths_per_s'])"; }
run "b2560 l4" TPAMD_BUCKET=2560
run "b2048 l4" TPAMD_BUCKET=2048
run "b2560 l6 q8" TPAMD_BUCKET=2560 TPAMD_LANES=6 GPU_MAX_HW_QUEUES=8
run "b2048 l6 q8" TPAMD_BUCKET=2048 TPAMD_LANES=6 GPU_MAX_HW_QUEUES=8
run "b0 l3 q8" TPAMD_BUCKET=0 GPU_MAX_HW_QUEUES=8
run "b1024 l8 q8" TPAMD_BUCKET=1024 TPAMD_LANES=8 GPU_MAX_HW_QUEUES=8
