# round 4: vpt_tokenize_batch, host loop as a stage of its own -- offsets cut per chunk, copies in two chunks ahead, copies out as soon as an
# earlier chunk's event has fired (r04_i: all six copies out started behind the last kernel).  Parity, A/B, chunk sizes, stress, timeline.
O=gpurun_out/r04_j; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -x -k "tokenize or predict_and_write or flat_kernel or pinned" 2>&1 | tail -3 | tee $O/tests.log
for E in "" "VPT_TOKENIZE_SERIAL=1" "VPT_TOKENIZE_CHUNK_BYTES=2400000" "VPT_TOKENIZE_CHUNK_BYTES=4800000" "VPT_TOKENIZE_CHUNK_BYTES=6400000" "VPT_TOKENIZE_CHUNK_BYTES=9600000" "VPT_TOKENIZE_DIRECT=1"; do
  env $E python tools/tokenize_bench.py 2>&1 | grep -v amdgpu.ids | tail -1 >> $O/tokenize.jsonl
done
for E in "" "VPT_TOKENIZE_SERIAL=1" "VPT_TOKENIZE_CHUNK_BYTES=16000000" "VPT_TOKENIZE_CHUNK_BYTES=8000000"; do
  env $E python tools/tokenize_bench.py --repeat 10 --iters 7 2>&1 | grep -v amdgpu.ids | tail -1 >> $O/tokenize.jsonl
done
cat $O/tokenize.jsonl | cut -c1-400
python tools/tokenize_stress.py --iters 250 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-500 > $O/stress.json; cat $O/stress.json
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OLDPWD/$O/trace -- python $OLDPWD/tools/tokenize_bench.py --iters 6 > $OLDPWD/$O/trace.log 2>&1
cd $OLDPWD
python tools/timeline.py $O/trace $O/timeline.txt 48
rm -rf $O/trace
