#!/bin/bash
# k_tower_halo with other s_setprio turn periods (conv_mainloop.cuh: TG_PRIO_PERIOD, TG_NO_PRIO_TURNS): libtakgpu variants timed under rocprofv3
#   bash scripts/probes/tower_prio_probe.sh "" "-DTG_PRIO_PERIOD=6" "-DTG_NO_PRIO_TURNS" ...
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
for d in "$@"; do
  EXTRA_DEFS="$d" bash $R/scripts/probes/fc_ring_probe.sh "0" 2>&1 | grep "^mask" | sed "s/^mask 0/[$d]/; s/ | k_softmax.*//"
done
