cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
run() { tag="$1"; shift; env "$@" timeout -k 10 300 python tools/gpu_other_configs.py 4 2>>gpurun_out/r03_f.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tag', d['buckets'], d['ms_per_batch'], d['paths_per_s'], d['buckets_equal_dof_groups_on_every_path'], d['bit_exact_on_sample'])"; }
run "chain3" TPAMD_BUCKET=0 TPAMD_CHAIN_FRONTS=3
run "chain2" TPAMD_BUCKET=0 TPAMD_CHAIN_FRONTS=2
run "chain3 tpbr128" TPAMD_BUCKET=0 TPAMD_CHAIN_FRONTS=3 TPAMD_K1_TPB_RAGGED=128
run "chain3 tpbr256" TPAMD_BUCKET=0 TPAMD_CHAIN_FRONTS=3 TPAMD_K1_TPB_RAGGED=256
TPAMD_CHAIN_FRONTS=3 TPAMD_BUCKET=0 timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/r03_f_tl -o b0 --output-format csv -- python3 tools/gpu_other_configs.py 4 > gpurun_out/r03_f_tl.log 2>&1
python tools/kernel_timeline.py gpurun_out/r03_f_tl/b0_kernel_trace.csv 12
