cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for cf in 2 1 0; do
TPAMD_CHAIN_FRONTS=$cf TPAMD_BUCKET=0 timeout -k 10 300 python tools/gpu_other_configs.py 4 2>>gpurun_out/r03_d.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('chain=$cf bucket=0', d['ms_per_batch'], d['paths_per_s'], d['dof_groups_in_turn_ms'], d['buckets_equal_dof_groups_on_every_path'], d['bit_exact_on_sample'])"
done
TPAMD_LANE_PRIORITY=0 TPAMD_BUCKET=0 timeout -k 10 300 python tools/gpu_other_configs.py 4 2>>gpurun_out/r03_d.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('chain=2 noprio', d['ms_per_batch'], d['paths_per_s'])"
TPAMD_BUCKET=0 timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/r03_d_tl -o b0 --output-format csv -- python3 tools/gpu_other_configs.py 4 > gpurun_out/r03_d_tl.log 2>&1
python tools/kernel_timeline.py gpurun_out/r03_d_tl/b0_kernel_trace.csv 12
