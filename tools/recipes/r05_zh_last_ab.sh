# round 5, the last seconds of the GPU budget: the final front end (p6) against runs of 4 K chars (r4) and a 2^17-bit summary with 4 workgroups per CU (s17o8)
O=gpurun_out/r05_zh; mkdir -p $O; export TMPDIR=/tmp; REPO=$(pwd)
for V in p6 s17o8 r4; do
  cp tools/prebuilt/libvaporetto_$V.so vaporetto_amd/lib/libvaporetto_hip.so
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/trace_$V -- python $REPO/bench.py --config 4 --quick --steps 8 --warmup 2 --no-cpu-baseline --no-e2e --no-emit > $REPO/$O/trace_$V.log 2>&1)
  grep -h "tag_front_flat\|tag_pass_kernel" $(find $O/trace_$V -name "*kernel_stats.csv") | awk -F'",' -v v=$V '{split($2,a,","); print v, substr($1,1,60), a[1], a[3]}' | tee -a $O/tag_kernel_stats.txt
  rm -rf $O/trace_$V
done
