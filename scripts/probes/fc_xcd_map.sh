#!/bin/bash
# k_fc_ring with the XCD-aware (row block, column block) mapping (-DTG_FC_XCD_MAP) against the plain one: launch time (fc_ring_probe.sh)
# and memory-side traffic (TCC_EA0 counters, scripts/pmc_traffic.py) of `scripts/ab_forward.py c2`.   bash scripts/probes/fc_xcd_map.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/fc_xcd
mkdir -p $O
for v in plain xcd plain xcd; do
    if [ $v = xcd ]; then export EXTRA_DEFS=-DTG_FC_XCD_MAP; else export EXTRA_DEFS=; fi
    bash $R/scripts/probes/fc_ring_probe.sh "0" 2>&1 | tail -1 | sed "s/^/$v: /"
    cd /tmp && export TMPDIR=/tmp
    rm -rf $O/pmc_$v
    TAKGPU_LIB=$R/scripts/probes/_bin/libtakgpu_fc_p0.so rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_WRREQ --output-format csv -d $O/pmc_$v -o p -- python3 $R/scripts/ab_forward.py c2 > /dev/null 2> $O/pmc_$v.err
    python3 $R/scripts/pmc_traffic.py $(find $O/pmc_$v -name '*counter_collection.csv' | head -1) k_fc_ring 36836352 $O/traffic_$v.json | tail -3
    rm -rf $O/pmc_$v
done
