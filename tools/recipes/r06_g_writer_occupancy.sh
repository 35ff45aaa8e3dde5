# round 6: the tagged writer by occupancy (-DVPT_EMIT_TAG_OCC=4/5/6/8: launch bounds) and its timing ablations on the records (e1: suffix bytes not written, e2: no suffixes, e4: no marks)
O=gpurun_out/${VPT_OUT:-r06_g}; mkdir -p $O
python tools/tag_bench.py --variants ${VPT_VARIANTS:-new,o4,o5,o6,o8,e1,e2,e4,new} > $O/tag_bench.jsonl 2> $O/tag_bench.err; python - <<PY
import json
for l in open("$O/tag_bench.jsonl"):
    r = json.loads(l); print(r["variant"], r["write_tagged_ms"], r["tagged_text_parity"], r["write_untagged_ms"], r["fill_tags_records_ms"])
PY
tail -3 $O/tag_bench.err
