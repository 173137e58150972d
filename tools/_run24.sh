cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for lib in libtpamd.so libtpamd_noguess.so; do
TPAMD_LIBRARY=$PWD/x-edr-trajectory-planning_amd/csrc/$lib timeout -k 10 300 python bench.py --steps 300 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$lib piped', d['value'], d['ms_per_step'], d['roofline']['kernels']['k_sweep'])"
done; done
for lib in libtpamd.so libtpamd_noguess.so; do
TPAMD_LIBRARY=$PWD/x-edr-trajectory-planning_amd/csrc/$lib timeout -k 10 300 python bench.py --steps 300 --warmup 5 --no-cpu-baseline --no-pipeline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$lib unpiped', d['value'], d['ms_per_step'], d['roofline']['kernels']['k_sweep'])"
TPAMD_LIBRARY=$PWD/x-edr-trajectory-planning_amd/csrc/$lib timeout -k 10 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-pipeline --paths-per-gpu 8192 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$lib 8192 unpiped', d['value'], d['ms_per_step'], d['roofline']['kernels']['k_sweep'])"
done
