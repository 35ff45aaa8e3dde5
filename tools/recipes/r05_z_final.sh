# round 5, the evidence run on the final sources: the GPU parity suite, a fuzz over non-BMP alphabets and every window, the traffic counters of every workload
# (two PMC passes each: read-request size classes, WRITE_SIZE; + the full counter set for configs[1] and configs[4], whose step runs the tag kernels), a
# kernel trace of configs[4] WITH the writer (every kernel of the tagged pipeline in one summary), the whole bench, the dry runs of the scaling jobs
O=gpurun_out/r05_z2; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > $O/gpu_tests.log; tail -2 $O/gpu_tests.log
( timeout 300 python tools/fuzz_gpu.py 200 2>&1 | grep -v amdgpu.ids | tail -2 ) > $O/fuzz_gpu.log; cat $O/fuzz_gpu.log
./tools/profile.sh r05_z2_c1 --config 1 > $O/profile_c1.log 2>&1; grep "traffic entry" $O/profile_c1.log | cut -c1-200
./tools/profile.sh r05_z2_c4 --config 4 > $O/profile_c4.log 2>&1; grep "traffic entry" $O/profile_c4.log | cut -c1-200
export VPT_PMC_GROUPS="TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum|WRITE_SIZE"
for C in 3 5 6 7 8 2; do
  ./tools/profile.sh r05_z2_c$C --config $C > $O/profile_c$C.log 2>&1
  grep "traffic entry" $O/profile_c$C.log | cut -c1-200
done
unset VPT_PMC_GROUPS
export TMPDIR=/tmp; REPO=$(pwd)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/trace_c4_emit -- python $REPO/bench.py --config 4 --quick --steps 5 --warmup 1 --no-cpu-baseline --no-e2e > $REPO/$O/trace_c4_emit.log 2>&1)
cat $(find $O/trace_c4_emit -name "*kernel_stats.csv" | head -1) > $O/c4_with_writer_kernel_stats.csv; head -12 $O/c4_with_writer_kernel_stats.csv | cut -c1-180
