set -e
cd /root/repo
for c in c2 c3 c5; do
  timeout -k 10 200 python scripts/ab_bits.py $c 4096
  TG_NO_HALO_TOWER=1 timeout -k 10 200 python scripts/ab_bits.py $c 4096
  TG_PRECISION=bf16x3 timeout -k 10 200 python scripts/ab_bits.py $c 4096
  TG_PRECISION=bf16x3 TG_NO_HALO_TOWER=1 timeout -k 10 200 python scripts/ab_bits.py $c 4096
done > gpurun_out/bits.log 2>&1
for c in c2 c3 c5; do
  timeout -k 10 200 python scripts/ab_forward.py $c
  TG_PRECISION=bf16x3 timeout -k 10 200 python scripts/ab_forward.py $c
done > gpurun_out/ab.log 2>&1
cd scripts/probes && timeout -k 10 120 ./_bin/tower_stamps 80 > ../../gpurun_out/f32_stamps.log 2>&1
for c in c2 c5 c3; do timeout -k 10 120 ./_bin/tower_s3_stamps $c; done > ../../gpurun_out/s3_stamps.log 2>&1
