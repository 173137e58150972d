cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for lib in libtpamd.so libtpamd_max_ilp.so libtpamd_iterative_ilp.so; do
TPAMD_LIBRARY=$PWD/x-edr-trajectory-planning_amd/csrc/$lib timeout -k 10 300 python bench.py --steps 300 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$lib piped', d['value'], d['ms_per_step'], d['config']['solved_paths'], {k:v['ms'] for k,v in d['roofline']['kernels'].items()})"
TPAMD_LIBRARY=$PWD/x-edr-trajectory-planning_amd/csrc/$lib timeout -k 10 300 python bench.py --steps 300 --warmup 5 --no-cpu-baseline --no-pipeline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$lib unpiped', d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['roofline']['kernels'].items()})"
done; done
