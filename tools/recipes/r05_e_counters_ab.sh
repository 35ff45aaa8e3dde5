# round 5: the full counter set of tools/profile.sh for the round-4 kernel (r05base) and the lighter one with the same table hashes (r05e2), configs[1]:
# what got worse while the instruction counts went down?
O=gpurun_out/r05_e; mkdir -p $O
for V in r05base r05e2; do
  cp tools/prebuilt/libvaporetto_$V.so vaporetto_amd/lib/libvaporetto_hip.so
  ./tools/profile.sh r05_e_$V --config 1 > $O/profile_$V.log 2>&1
  cp gpurun_out/prof_r05_e_$V/summary.txt $O/summary_$V.txt
  grep -c avg $O/summary_$V.txt
done
