# round 5: the parity test of the front end on batches whose labels run out (one-char sentences), on the device
O=gpurun_out/r05_zi; mkdir -p $O
( timeout 18 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_fill_tags_front_end_where_the_labels_run_out" 2>&1 | tail -3 ) | tee $O/gpu_labels_run_out.log
