# round 5: the flat front end of fill_tags, variants named on the command line (LIB or LIB:ENV=VAL): kernel stats of configs[4] step; a *p library also prints its clock shares
O=gpurun_out/r05_zc; mkdir -p $O; export TMPDIR=/tmp; REPO=$(pwd)
one() {   # tag, library, env...
  T=$1; V=$2; shift 2
  cp tools/prebuilt/libvaporetto_$V.so vaporetto_amd/lib/libvaporetto_hip.so
  (cd /tmp && env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/trace_$T -- python $REPO/bench.py --config 4 --quick --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-emit > $REPO/$O/trace_$T.log 2>&1)
  python - "$T" $(find $O/trace_$T -name "*kernel_stats.csv") <<'PY' | tee -a $O/tag_kernel_stats.txt
import csv, sys
tag = sys.argv[1]
for row in csv.DictReader(open(sys.argv[2])):
    n = row["Name"]
    if "tag_front" in n or "tag_pass" in n:
        print("%-8s %-44s calls %3s  avg %9.1f us" % (tag, n.replace("vpt::(anonymous namespace)::", "")[:44], row["Calls"], float(row["AverageNs"]) / 1e3))
PY
  rm -rf $O/trace_$T
}
for V in "$@"; do L=${V%%:*}; E=${V#*:}; [ "$E" = "$V" ] && E=A=1; one $(echo $V | tr ":=" "__") $L $E; done
for V in "$@"; do case $V in *p)
  cp tools/prebuilt/libvaporetto_$V.so vaporetto_amd/lib/libvaporetto_hip.so
  python bench.py --config 4 --quick --steps 4 --warmup 1 --no-cpu-baseline --no-e2e --no-emit > $O/bench_$V.json 2> $O/bench_$V.err
  grep "tag front profile" $O/bench_$V.err | tail -1 | tee $O/tag_front_profile_$V.txt;; esac; done
