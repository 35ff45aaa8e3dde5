# round 5: the flat front end of fill_tags with a third fewer vector instructions per step (t6 / t8: compiled for 6 / 8 waves per SIMD; t6r4 / t6r1: runs of
# 4 K / 1 K chars; WGS_PER_CU: the grid) against the round-4 kernel (t0): kernel stats of configs[4]'s step (rocprofv3 --kernel-trace --stats), then the tags of
# the built library against the oracle (bench parity + the GPU tag tests)
O=gpurun_out/r05_u; mkdir -p $O; export TMPDIR=/tmp; REPO=$(pwd)
cp vaporetto_amd/lib/libvaporetto_hip.so /tmp/lib_built.so
one() {   # tag, library, env...
  T=$1; V=$2; shift 2
  cp tools/prebuilt/libvaporetto_$V.so vaporetto_amd/lib/libvaporetto_hip.so
  (cd /tmp && env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$O/trace_$T -- python $REPO/bench.py --config 4 --quick --steps 10 --warmup 2 --no-cpu-baseline --no-e2e --no-emit > $REPO/$O/trace_$T.log 2>&1)
  echo "== $T" | tee -a $O/tag_kernel_stats.txt
  grep -h "tag_front_flat\|tag_pass_kernel\|tag_tokens" $(find $O/trace_$T -name "*kernel_stats.csv") | cut -d, -f1-4 | sed 's/vpt::(anonymous namespace):://' | tee -a $O/tag_kernel_stats.txt
  rm -rf $O/trace_$T
}
one t0 t0 A=1
one t6 t6 A=1
one t8 t8 A=1
one t6r4 t6r4 A=1
one t6r1 t6r1 A=1
one t6_wg8 t6 VPT_TAG_WGS_PER_CU=8
one t6_wg16 t6 VPT_TAG_WGS_PER_CU=16
one t6_wg64 t6 VPT_TAG_WGS_PER_CU=64
one t8_wg16 t8 VPT_TAG_WGS_PER_CU=16
one t0b t0 A=1
cp /tmp/lib_built.so vaporetto_amd/lib/libvaporetto_hip.so
python bench.py --config 4 --quick --steps 5 --warmup 2 --no-e2e > $O/bench_c4.json 2> $O/bench_c4.err; python -c "
import json; l=json.loads(open('$O/bench_c4.json').read().strip().splitlines()[-1]); print('configs[4]', l['ms_per_step'], l['parity'], l['tags']['ms_per_step'], l['tags']['parity'], l['emit']['ms_per_step'], l['emit']['parity'])"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tag or fill or tokenize" 2>&1 | tail -3 | tee $O/pytest_tags.txt
