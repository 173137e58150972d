cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for seed in 351 352 353; do
timeout -k 10 420 python tools/gpu_fuzz.py 300 $seed > gpurun_out/r03_fuzz_$seed.json 2> gpurun_out/r03_fuzz_$seed.err; echo rc=$?; cut -c1-160 gpurun_out/r03_fuzz_$seed.json
done
