# round 4, the evidence run on the final sources: parity suite, a long fuzz over every window and the fused writer, the traffic counters of every
# workload (two PMC passes each: read-request size classes, WRITE_SIZE; + the full counter set for configs[1]), the dry run of the scaling job
O=gpurun_out/r04_y; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > $O/gpu_tests.log; tail -2 $O/gpu_tests.log
( timeout 420 python tools/fuzz_gpu.py 330 2>&1 | grep -v amdgpu.ids | tail -2 ) > $O/fuzz_gpu.log; cat $O/fuzz_gpu.log
( VPT_FUZZ_SEED0=50000 timeout 300 python tools/fuzz_gpu.py 200 2>&1 | grep -v amdgpu.ids | tail -2 ) >> $O/fuzz_gpu.log; tail -1 $O/fuzz_gpu.log
./tools/profile.sh r04_y_c1 --config 1 > $O/profile_c1.log 2>&1
export VPT_PMC_GROUPS="TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum|WRITE_SIZE"
for C in 3 4 5 6 7 2; do
  ./tools/profile.sh r04_y_c$C --config $C > $O/profile_c$C.log 2>&1
  grep "traffic entry" $O/profile_c$C.log | cut -c1-260
done
unset VPT_PMC_GROUPS
python bench.py --dry-scale --steps 10 --warmup 2 --no-e2e --no-emit > $O/dry_scale.jsonl 2> $O/dry_scale.err; tail -1 $O/dry_scale.err
