#!/bin/bash
# Second part of the round's evidence (gpurun's limit is 20 minutes per call): the training kernels' SQ counters, the reference's own
# constants (32 games x Net6 16x128) under the kernel trace, the games sweep and tg_policy_eval's small-batch latency, into gpurun_out/ev2/:
#   /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash scripts/collect_evidence_small_batch.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/ev2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P=python3
step() { echo "[evidence] $1 ($(date +%T))"; }
step "SQ counters, training kernels on one stream (two passes): MFMA busy, LDS bank conflicts of k_wgrad_halo / k_conv_halo"
TG_TRAIN_ONE_STREAM=1 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_train1 -o p -- python3 $R/scripts/train_step_ab.py 6 > /dev/null 2> $O/pmc_train.err
TG_TRAIN_ONE_STREAM=1 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_train2 -o p -- python3 $R/scripts/train_step_ab.py 6 > /dev/null 2>> $O/pmc_train.err
step "the reference's own constants (6x6, 32 games, Net6 16x128): kernel trace of two plies at 2000 rollouts, split towers and one workgroup per position"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_ref -o kt -- python3 $R/scripts/soak_selfplay.py --board 6 --games 32 --rollouts 2000 --plies 2 --every 1 --evaluator resnet --blocks 16 --filters 128 > $O/reference_constants_soak.log 2> $O/kt_ref.err
TG_NO_SPLIT_TOWER=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_ref0 -o kt -- python3 $R/scripts/soak_selfplay.py --board 6 --games 32 --rollouts 2000 --plies 2 --every 1 --evaluator resnet --blocks 16 --filters 128 > /dev/null 2>> $O/kt_ref.err
step "games sweep (C2 network, 32 ... 16384 games) and tg_policy_eval latency at small batches, both tower forms"
python3 $R/scripts/games_sweep.py > $O/games_sweep_c2.jsonl 2> $O/sweep.err
python3 $R/scripts/policy_eval_latency.py 2>> $O/sweep.err | grep -v amdgpu.ids > $O/policy_eval_latency_split_tower.txt
TG_NO_SPLIT_TOWER=1 python3 $R/scripts/policy_eval_latency.py 2>> $O/sweep.err | grep -v amdgpu.ids > $O/policy_eval_latency_one_workgroup_per_position.txt
step "summaries"
for d in pmc_train1 pmc_train2; do $P $R/scripts/pmc_summary.py $(find $O/$d -name '*counter_collection.csv' | head -1) k_conv_halo k_wgrad_halo k_bn_bwd_apply k_bn_fwd_apply k_wgrad_reduce_conv; done > $O/pmc_sq_train_kernels.txt
cp $(find $O/kt_ref -name '*kernel_stats.csv' | head -1) $O/kernel_stats_reference_constants_split_tower.csv 2>/dev/null
cp $(find $O/kt_ref0 -name '*kernel_stats.csv' | head -1) $O/kernel_stats_reference_constants_one_workgroup_per_position.csv 2>/dev/null
rm -rf $O/kt_ref $O/kt_ref0 $O/pmc_train1 $O/pmc_train2
ls -la $O
step "done"
