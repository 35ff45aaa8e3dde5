# round 4: vpt_predict_batch with a stream per direction of the link (VPT_PIPE_DUPLEX=1) against the four in-order lanes (default), chunk sizes; the
# labels-only call; ten batches as one call (default there: the event pipeline).  Parity in every line.
O=gpurun_out/r04_s; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -x -k "pipelined_host_path or labels_only" 2>&1 | tail -2 | tee $O/tests.log
for E in "X=1" "VPT_PIPE_DUPLEX=1" "VPT_PIPE_DUPLEX=1 VPT_CHUNK_CHARS=1100000" "VPT_PIPE_DUPLEX=1 VPT_CHUNK_CHARS=1650000" "VPT_PIPE_DUPLEX=1 VPT_CHUNK_CHARS=2200000" "VPT_PIPE_DUPLEX=1 VPT_CHUNK_CHARS=820000"; do
  env $E python tools/e2e_bench.py 2>&1 | grep -v amdgpu.ids | tail -1 >> $O/e2e.jsonl
done
for E in "X=1" "VPT_PIPE_DUPLEX=1 VPT_CHUNK_CHARS=1100000" "VPT_PIPE_DUPLEX=1 VPT_CHUNK_CHARS=1650000"; do
  env $E python tools/e2e_bench.py --labels-only 2>&1 | grep -v amdgpu.ids | tail -1 >> $O/e2e.jsonl
done
for E in "X=1" "VPT_PIPE_DUPLEX=1 VPT_CHUNK_CHARS=2200000" "VPT_PIPE_DUPLEX=1 VPT_CHUNK_CHARS=4400000"; do
  env $E python tools/e2e_bench.py --repeat 10 --iters 7 2>&1 | grep -v amdgpu.ids | tail -1 >> $O/e2e.jsonl
done
cat $O/e2e.jsonl | cut -c1-330
