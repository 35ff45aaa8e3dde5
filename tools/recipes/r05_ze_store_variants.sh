# round 5: the new parity test of the front end's store variants (1 .. 5 tag slots, aligned and unaligned tag arrays) on the device
O=gpurun_out/r05_ze; mkdir -p $O
( timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_fill_tags_front_end_stores or test_tag_front_end_over_runs or test_fill_tags_as_two" 2>&1 | tail -4 ) | tee $O/gpu_store_variants.log
