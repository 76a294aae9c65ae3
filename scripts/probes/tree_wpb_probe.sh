#!/bin/bash
# k_backup_select with other block sizes (TG_WPB waves = games per block): libtakgpu variants of search_kernels.hip timed inside a short bench under rocprofv3
#   bash scripts/probes/tree_wpb_probe.sh 4 1 2 8
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
B=$R/scripts/probes/_bin; O=$R/gpurun_out/wpb; mkdir -p $B $O
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-result"
OBJS=$(ls $R/tak_amd/csrc/_obj/*.o | grep -v search_kernels.o)
cd /tmp; export TMPDIR=/tmp
for w in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -DTG_WPB=$w -c $R/tak_amd/csrc/search_kernels.hip -o $B/search_kernels_w$w.o || exit 1
  /opt/rocm/bin/hipcc $FLAGS -shared -o $B/libtakgpu_w$w.so $OBJS $B/search_kernels_w$w.o -ldl || exit 1
  export TAKGPU_LIB=$B/libtakgpu_w$w.so
  rm -rf $O/kt_$w
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$w -o kt -- python3 $R/bench.py --no-extras --no-cpu-baseline --no-alt-precision --no-train --steps 2 --warmup 1 > $O/b_$w.json 2> $O/kt_$w.err
  python3 - "$w" "$(find $O/kt_$w -name '*kernel_stats.csv' | head -1)" "$O/b_$w.json" <<'PY'
import csv, json, sys
rows = {r["Name"].split("(")[0].replace("void ", "").replace("tg::", "")[:24]: r for r in csv.DictReader(open(sys.argv[2]))}
print("WPB", sys.argv[1], " | ".join(f"{k}: {float(r['AverageNs']) / 1e3:.1f} us" for k, r in rows.items() if k.startswith(("k_backup_select", "k_fc_ring", "k_tower_halo"))), "| value", round(json.load(open(sys.argv[3]))["value"]))
PY
  rm -rf $O/kt_$w
done
