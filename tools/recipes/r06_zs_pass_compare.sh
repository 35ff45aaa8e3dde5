# round 6: tag_pass_kernel's n-gram compare with the record's symbols as fields of registers known at compile time (unrolled over the wave's step) instead of a 192-bit
# shift register moved every step -- `new` against `head`, same box, twice; the kernels' times by rocprofv3; the tag tests
O=gpurun_out/r06_zs; mkdir -p $O
for R in 1 2; do python tools/tag_bench.py --variants head,new 2>>$O/tag.err | tee -a $O/tag_bench.jsonl | cut -c1-300; done
( timeout 900 python -m pytest tests -m gpu -x -q -n 4 -k "tag" 2>&1 | tail -4 ) > $O/gpu_tag_tests.log; tail -2 $O/gpu_tag_tests.log
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python $GRAFT_REPO_ROOT/tools/tag_bench.py --variants head,new --steps 5 > $GRAFT_REPO_ROOT/$O/trace.log 2>&1
cd $GRAFT_REPO_ROOT; python - <<'PY' | tee gpurun_out/r06_zs/pass_kernel_times.txt
import glob, csv
f = glob.glob("gpurun_out/r06_zs/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "tag_pass_kernel" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
h = len(d) // 2
print("tag_pass_kernel ms, launches of the first library (head) then the second (new): median %.4f / %.4f  (n = %d / %d)" % (sorted(d[:h])[h // 2], sorted(d[h:])[(len(d) - h) // 2], h, len(d) - h))
PY
rm -rf $O/trace
