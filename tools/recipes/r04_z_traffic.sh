# round 4, last call: the traffic counters of every workload's scoring kernel on the final sources (bench with --no-emit: only the plain instance of
# the kernel runs), the full counter set for configs[1], kernel-trace summaries, and the bench line of every workload
O=gpurun_out/r04_z; mkdir -p $O
./tools/profile.sh r04_z_c1 --config 1 > $O/profile_c1.log 2>&1; grep "traffic entry" $O/profile_c1.log | cut -c1-260
export VPT_PMC_GROUPS="TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum|WRITE_SIZE"
for C in 3 4 5 6 7 2; do
  ./tools/profile.sh r04_z_c$C --config $C > $O/profile_c$C.log 2>&1
  grep "traffic entry" $O/profile_c$C.log | cut -c1-260
done
unset VPT_PMC_GROUPS
python bench.py --steps 20 --warmup 3 > $O/bench_all.json 2> $O/bench_all.err; tail -1 $O/bench_all.err | cut -c1-200
