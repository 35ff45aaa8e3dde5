# round 5: what the memory system gives a kernel with the traffic of fill_tags' front end and nothing else (tools/stream_bench.hip)
O=gpurun_out/r05_w; mkdir -p $O
timeout 120 tools/stream_bench 133000000 | tee $O/stream_bench.txt
