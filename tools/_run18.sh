cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for lib in libtpamd.so libtpamd_t16.so; do
TPAMD_LIBRARY=$PWD/x-edr-trajectory-planning_amd/csrc/$lib timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'], d['roofline']['kernels'], d['config']['solved_paths'])"
done
timeout -k 10 400 python tools/gpu_timeline.py r03_q_tl x-edr-trajectory-planning_amd/csrc/libtpamd.so x-edr-trajectory-planning_amd/csrc/libtpamd_t16.so 2>&1 | tail -2
