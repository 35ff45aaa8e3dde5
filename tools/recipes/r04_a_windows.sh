# round 4, first call: the packed tables for every window on the GPU -- parity suite, charw4 before (general kernel, forced) and
# after (specialised kernel, row window 4) with counters, then every workload of the bench
O=gpurun_out/r04_a; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > $O/gpu_tests.log
VPT_FORCE_GENERIC=1 python bench.py --config 7 --steps 20 --warmup 3 --no-e2e --no-emit 2>&1 | grep -v amdgpu.ids | tail -1 > $O/charw4_general.json
python bench.py --config 7 --steps 20 --warmup 3 --no-e2e --no-emit 2>&1 | grep -v amdgpu.ids | tail -1 > $O/charw4_packed.json
VPT_FORCE_GENERIC=1 ./tools/profile.sh r04_a_charw4_general --config 7 > $O/profile_general.log 2>&1
./tools/profile.sh r04_a_charw4_packed --config 7 > $O/profile_packed.log 2>&1
python bench.py --steps 20 --warmup 3 2>&1 | grep -v amdgpu.ids | tail -1 > $O/bench_all.json
