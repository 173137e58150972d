cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for dl in 0 3 6 10 15 25 40; do
TPAMD_FRONT_DELAY_US=$dl timeout -k 10 300 python bench.py --steps 300 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('delay $dl piped', d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['roofline']['kernels'].items()})"
done; done
TPAMD_LIBRARY=$PWD/x-edr-trajectory-planning_amd/csrc/libtpamd_prev.so timeout -k 10 300 python bench.py --steps 300 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('prev piped', d['value'], d['ms_per_step'])"
