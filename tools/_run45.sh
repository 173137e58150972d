cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for args in "--steps 20 --warmup 5 --timing-stride 20" "--steps 20 --warmup 5 --timing-stride 4" "--steps 20 --warmup 5 --no-kernel-timing"; do
TPAMD_BENCH_TIMING_OFFSET=3 timeout -k 10 300 python bench.py $args --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$args', d['value'], d['ms_per_step'], 'cold', d.get('cold_value'))"
done; done
