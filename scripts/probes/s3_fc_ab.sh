#!/bin/bash
# k_fc_s3_ring (split-bf16 policy FC, LDS-DMA ring) with parts removed / other wave counts: builds libtakgpu variants of
# net_s3_kernels.hip with the given -D flags and times the FC under rocprofv3 (scripts/ab_forward.py c2, TG_PRECISION=bf16x3).
#   bash scripts/probes/s3_fc_ab.sh "" "-DTG_FSR_NW=4" ...      ("" = the product build; the TG_FSR_PROBE masks of round 4 lived in commit 6da8f61)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
B=$R/scripts/probes/_bin; O=$R/gpurun_out/s3ab; mkdir -p $B $O
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-result"
OBJS=$(ls $R/tak_amd/csrc/_obj/*.o | grep -v net_s3_kernels.o)
cd /tmp; export TMPDIR=/tmp; export TG_PRECISION=bf16x3
i=0
for defs in "$@"; do
  i=$((i+1))
  unset TAKGPU_LIB
  if [ -n "$defs" ]; then
    /opt/rocm/bin/hipcc $FLAGS $defs -c $R/tak_amd/csrc/net_s3_kernels.hip -o $B/net_s3_v$i.o || exit 1
    /opt/rocm/bin/hipcc $FLAGS -shared -o $B/libtakgpu_s3_v$i.so $OBJS $B/net_s3_v$i.o -ldl || exit 1
    export TAKGPU_LIB=$B/libtakgpu_s3_v$i.so
  fi
  rm -rf $O/kt_$i
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$i -o kt -- python3 $R/scripts/ab_forward.py c2 > $O/ab_$i.json 2> $O/kt_$i.err
  python3 - "[$defs]" "$(find $O/kt_$i -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv, sys
rows = {r["Name"].split("(")[0].replace("void ", "").replace("tg::", "")[:24]: r for r in csv.DictReader(open(sys.argv[2]))}
print(sys.argv[1], " | ".join(f"{k}: {float(r['AverageNs']) / 1e3:.1f} us" for k, r in rows.items() if k.startswith(("k_fc", "k_tower_s3", "k_softmax"))))
PY
  rm -rf $O/kt_$i
done
