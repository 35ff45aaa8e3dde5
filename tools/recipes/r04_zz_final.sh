# round 4, the evidence run on the final sources (after the tokenize pipeline's rework and phase D's use of phase A's scan): parity suite, fuzz over every
# window and the fused writer, traffic counters of every workload's plain scoring kernel, the bench line of every workload
O=gpurun_out/r04_zz; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > $O/gpu_tests.log; tail -2 $O/gpu_tests.log
( timeout 200 python tools/fuzz_gpu.py 150 2>&1 | grep -v amdgpu.ids | tail -2 ) > $O/fuzz_gpu.log; cat $O/fuzz_gpu.log
export VPT_PMC_GROUPS="TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum|WRITE_SIZE"
for C in 1 3 4 5 6 7 2; do
  ./tools/profile.sh r04_zz_c$C --config $C > $O/profile_c$C.log 2>&1
  grep "traffic entry" $O/profile_c$C.log | cut -c1-260
done
unset VPT_PMC_GROUPS
python bench.py --steps 20 --warmup 3 > $O/bench_all.json 2> $O/bench_all.err; tail -1 $O/bench_all.err | cut -c1-200
