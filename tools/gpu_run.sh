#!/bin/bash
# HBM-side traffic of the scoring kernel on the final sources (read-request size classes + WRITE_SIZE, separate passes) -> profiles/traffic.json
O=gpurun_out/r03_zt; mkdir -p $O
export TMPDIR=/tmp
G="TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum|WRITE_SIZE"
VPT_PMC_GROUPS="$G" ./tools/profile.sh r03_zt_c1 --config 1 > $O/profile_c1.log 2>&1; grep "traffic entry" $O/profile_c1.log | cut -c1-330
VPT_PMC_GROUPS="$G" ./tools/profile.sh r03_zt_c2 > $O/profile_c2.log 2>&1; grep "traffic entry" $O/profile_c2.log | cut -c1-330
VPT_PMC_GROUPS="$G" ./tools/profile.sh r03_zt_c3 --config 3 > $O/profile_c3.log 2>&1; grep "traffic entry" $O/profile_c3.log | cut -c1-330
