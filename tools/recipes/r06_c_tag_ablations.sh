# round 6: where the time of fill_tags (records) and of the tagged writer goes after the dense arrays are gone -- A/B of prebuilt ablation libraries in one process
# (tools/build_variants.sh e1:-DVPT_EMIT_ABLATE=1 e2:..=2 e4:..=4 t1:-DVPT_TAG_ABLATE=1 t2:..=2 t8:..=8 t64:..=64)
O=gpurun_out/r06_c; mkdir -p $O
python tools/tag_bench.py --variants new,e1,e2,e4,t1,t2,t8,t64,new > $O/tag_bench.jsonl 2> $O/tag_bench.err; cat $O/tag_bench.jsonl | cut -c1-600; tail -3 $O/tag_bench.err
