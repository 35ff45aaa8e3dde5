# round 5: the tag kernels reading TagParams through the kernarg pointer (tg1), + a fence per step of the front end (tg2), against by value (w0): kernel stats of
# configs[4]'s step (rocprofv3 --kernel-trace --stats), tags checked against the oracle in a separate parity run of the last one
O=gpurun_out/r05_o; mkdir -p $O; export TMPDIR=/tmp; REPO=$(pwd)
for V in r05w0 r05tg1 r05tg2; do
  cp tools/prebuilt/libvaporetto_$V.so vaporetto_amd/lib/libvaporetto_hip.so
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/trace_$V -- python $REPO/bench.py --config 4 --quick --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-emit > $REPO/$O/trace_$V.log 2>&1)
  echo "== $V" | tee -a $O/tag_kernel_stats.txt
  grep -h "tag_front_flat\|tag_pass_kernel\|tag_tokens" $(find $O/trace_$V -name "*kernel_stats.csv") | cut -d, -f1-4 | sed 's/vpt::(anonymous namespace):://' | tee -a $O/tag_kernel_stats.txt
done
python bench.py --config 4 --quick --steps 5 --warmup 2 --no-e2e > $O/bench_c4.json 2> $O/bench_c4.err; python -c "
import json; l=json.loads(open('$O/bench_c4.json').read().strip().splitlines()[-1]); print('configs[4]', l['ms_per_step'], l['parity'], l['tags']['ms_per_step'], l['tags']['parity'], l['emit']['ms_per_step'], l['emit']['parity'])"
