# round 4: vpt_tokenize_batch with scoring (labels only) and the writer as launches of their own per chunk (VPT_TOKENIZE_SCHEDULE=3), chunks placed
# by an upper bound and closed up by the copies out, against the fused + chained schedule (1).  Parity, A/B, chunk sizes, stress, timeline.
O=gpurun_out/r04_m; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -x -k "tokenize or predict_and_write or flat_kernel or pinned" 2>&1 | tail -3 | tee $O/tests.log
for E in "" "VPT_TOKENIZE_SCHEDULE=1" "VPT_TOKENIZE_CHUNK_BYTES=1600000" "VPT_TOKENIZE_CHUNK_BYTES=2400000" "VPT_TOKENIZE_CHUNK_BYTES=4800000" "VPT_TOKENIZE_CHUNK_BYTES=6400000"; do
  env $E python tools/tokenize_bench.py 2>&1 | grep -v amdgpu.ids | tail -1 >> $O/tokenize.jsonl
done
for E in "" "VPT_TOKENIZE_SCHEDULE=1" "VPT_TOKENIZE_CHUNK_BYTES=4000000" "VPT_TOKENIZE_CHUNK_BYTES=16000000"; do
  env $E python tools/tokenize_bench.py --repeat 10 --iters 7 2>&1 | grep -v amdgpu.ids | tail -1 >> $O/tokenize.jsonl
done
cat $O/tokenize.jsonl | cut -c1-400
python tools/tokenize_stress.py --iters 250 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-500 > $O/stress.json; cat $O/stress.json
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OLDPWD/$O/trace -- python $OLDPWD/tools/tokenize_bench.py --iters 6 > $OLDPWD/$O/trace.log 2>&1
cd $OLDPWD
python tools/timeline.py $O/trace $O/timeline.txt 56
rm -rf $O/trace
