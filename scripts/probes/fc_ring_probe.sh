#!/bin/bash
# Library variants of net_kernels.hip built with -D flags (EXTRA_DEFS / PROBE_MACRO=value; round 4 used the TG_RING_PROBE masks that
# commit 6da8f61 still carried — removed from the product in round 5): what each candidate change of the policy FC is worth.  Builds one libtakgpu per mask into scripts/probes/_bin/, runs scripts/ab_forward.py c2 on each under
# rocprofv3 --kernel-trace --stats and prints the average k_fc_ring launch.  Run on the GPU box:
#   bash scripts/probes/fc_ring_probe.sh "0 64 128 192"
set -u
EXTRA_DEFS=${EXTRA_DEFS:-}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
B=$R/scripts/probes/_bin
O=$R/gpurun_out/fc_probe
mkdir -p $B $O
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-result"
OBJS=$(ls $R/tak_amd/csrc/_obj/*.o | grep -v net_kernels.o)
cd /tmp && export TMPDIR=/tmp
for m in ${1:-0 64 128 192}; do
    /opt/rocm/bin/hipcc $FLAGS -D${PROBE_MACRO:-TG_RING_PROBE}=$m $EXTRA_DEFS -c $R/tak_amd/csrc/net_kernels.hip -o $B/net_kernels_p$m.o || exit 1
    /opt/rocm/bin/hipcc $FLAGS -shared -o $B/libtakgpu_fc_p$m.so $OBJS $B/net_kernels_p$m.o -ldl || exit 1
    export TAKGPU_LIB=$B/libtakgpu_fc_p$m.so
    rm -rf $O/kt_$m
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$m -o kt -- python3 $R/scripts/ab_forward.py c2 > $O/ab_$m.json 2> $O/kt_$m.err
    f=$(find $O/kt_$m -name '*kernel_stats.csv' | head -1)
    python3 - "$m" "$f" <<'PY' | tee -a $O/summary.txt
import csv, sys
rows = {r["Name"].split("(")[0].replace("void ", "").replace("tg::", "")[:28]: r for r in csv.DictReader(open(sys.argv[2]))}
print("mask", sys.argv[1], " | ".join(f"{k}: {float(r['AverageNs']) / 1e3:.1f} us x{r['Calls']}" for k, r in rows.items() if k.startswith(("k_fc", "k_tower_halo", "k_softmax"))))
PY
    rm -rf $O/kt_$m
done
