#!/bin/bash
# Collects the round's measurement evidence on a GPU box into gpurun_out/ev/ (copy what is to be judged into profiles/):
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash scripts/collect_evidence.sh'
# Every rocprofv3 invocation has the python program itself behind `--`; counters are collected in passes of their own.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/ev
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
step() { echo "[evidence] $1 ($(date +%T))"; }

step "bench.py, default flags"
( time python3 $R/bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_default.time
step "kernel trace of bench.py --no-extras --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_bench -o kt -- python3 $R/bench.py --no-extras --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/kt_bench.err
step "kernel trace of the training step: on one stream (the kernels' own durations), then as it runs (what overlaps)"
TG_TRAIN_ONE_STREAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_train -o kt -- python3 $R/scripts/train_step_ab.py 10 > $O/train_step_one_stream_under_rocprof.json 2> $O/kt_train.err
rocprofv3 --kernel-trace --output-format csv -d $O/kt_train2 -o kt -- python3 $R/scripts/train_step_ab.py 12 --driver > /dev/null 2> $O/kt_train2.err
python3 $R/scripts/probes/train_overlap.py $(find $O/kt_train2 -name '*kernel_trace.csv' | head -1) > $O/train_overlap_default.txt 2>&1
step "training step timings (no profiler): tg_train, tg_train_chunk, one stream"
{ python3 $R/scripts/train_step_ab.py 40 --driver; python3 $R/scripts/train_step_ab.py 40; TG_TRAIN_ONE_STREAM=1 python3 $R/scripts/train_step_ab.py 40 --driver; } > $O/train_step.jsonl 2>> $O/kt_train.err
step "training step's forward error budget per layer, and the full gradient audit (fp64 with the engine's / its own / PyTorch-f32's ReLU decisions)"
python3 $R/scripts/train_error_budget.py $O/train_error_budget.json > $O/train_error_budget.txt 2>&1
( cd $R && TG_C5_FULL_AUDIT=1 python3 -m pytest tests/test_gpu_c5_realsize.py -m gpu -x -q -s > $O/c5_realsize_full_audit.log 2>&1; cp gpurun_out/c5_realsize_gradient_parity.json $O/c5_realsize_gradient_parity_full_audit.json; cp gpurun_out/c5_realsize_layerwise_backward.json $O/ )
step "HBM traffic counters, forward C2"
rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_WRREQ --output-format csv -d $O/pmc_traffic -o p -- python3 $R/scripts/ab_forward.py c2 > $O/ab_forward_c2_under_pmc.json 2> $O/pmc_traffic.err
step "HBM traffic counters, policy FC inside the search loop (gather epilogue)"
rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_WRREQ --output-format csv -d $O/pmc_traffic_bench -o p -- python3 $R/bench.py --no-extras --no-cpu-baseline --no-alt-precision --no-train --steps 1 --warmup 1 > /dev/null 2> $O/pmc_traffic_bench.err
step "SQ counters, forward C2 (two passes)"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq1 -o p -- python3 $R/scripts/ab_forward.py c2 > /dev/null 2> $O/pmc_sq1.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_SALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq2 -o p -- python3 $R/scripts/ab_forward.py c2 > /dev/null 2> $O/pmc_sq2.err
step "SQ counters, tree kernels inside the bench"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_tree -o p -- python3 $R/bench.py --no-extras --no-cpu-baseline --no-alt-precision --no-train --steps 1 --warmup 1 > /dev/null 2> $O/pmc_tree.err
step "board pass micro-benchmark and its instruction mix"
python3 $R/scripts/bench_board.py > $O/board_pass_5x5.json 2> $O/board.err
python3 $R/scripts/bench_board.py --board 6 > $O/board_pass_6x6.json 2>> $O/board.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_board -o p -- python3 $R/scripts/bench_board.py --reps 3 > /dev/null 2>> $O/board.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_board6 -o p -- python3 $R/scripts/bench_board.py --board 6 --reps 3 > /dev/null 2>> $O/board.err
step "HBM traffic counters, board pass (5x5 and 6x6)"
rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_WRREQ --output-format csv -d $O/pmc_board_t5 -o p -- python3 $R/scripts/bench_board.py --reps 3 > /dev/null 2>> $O/board.err
rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_WRREQ --output-format csv -d $O/pmc_board_t6 -o p -- python3 $R/scripts/bench_board.py --board 6 --reps 3 > /dev/null 2>> $O/board.err
step "forward micro-benchmarks"
for c in c2 c3 c5; do python3 $R/scripts/ab_forward.py $c 2>/dev/null | grep cfg; done > $O/ab_forward.jsonl
step "summaries"
P=python3
$P $R/scripts/pmc_traffic.py $(find $O/pmc_traffic -name '*counter_collection.csv' | head -1) k_tower_halo 27262976 $O/pmc_traffic_k_tower_halo.json > /dev/null
$P $R/scripts/pmc_traffic.py $(find $O/pmc_traffic -name '*counter_collection.csv' | head -1) k_fc_ring 36836352 $O/pmc_traffic_k_fc_ring.json > /dev/null
# search loop: activations 26.2 MB + the 99 tiles' weights 10.1 MB + child_pidx 1.0 MB in, children's logits 0.7 MB + statistics 0.4 MB out
$P $R/scripts/pmc_traffic.py $(find $O/pmc_traffic_bench -name '*counter_collection.csv' | head -1) k_fc_ring 38563840 $O/pmc_traffic_k_fc_ring_gather.json > /dev/null
for d in pmc_sq1 pmc_sq2; do $P $R/scripts/pmc_summary.py $(find $O/$d -name '*counter_collection.csv' | head -1) k_tower_halo k_fc_ring; done > $O/pmc_sq_tower_fc.txt
$P $R/scripts/pmc_summary.py $(find $O/pmc_tree -name '*counter_collection.csv' | head -1) k_backup_select k_select k_reroot > $O/pmc_tree_kernels.txt
$P $R/scripts/pmc_summary.py $(find $O/pmc_board -name '*counter_collection.csv' | head -1) k_board_pass > $O/pmc_board_pass_instruction_mix.txt
$P $R/scripts/pmc_summary.py $(find $O/pmc_board6 -name '*counter_collection.csv' | head -1) k_board_pass >> $O/pmc_board_pass_instruction_mix.txt
# board pass: 2^20 positions x (state in + state out + unpadded f32 planes) = 7712 B (5x5) / 14016 B (6x6)
$P $R/scripts/pmc_traffic.py $(find $O/pmc_board_t5 -name '*counter_collection.csv' | head -1) k_board_pass 8086618112 $O/pmc_traffic_k_board_pass_5x5.json > /dev/null
$P $R/scripts/pmc_traffic.py $(find $O/pmc_board_t6 -name '*counter_collection.csv' | head -1) k_board_pass 14696841216 $O/pmc_traffic_k_board_pass_6x6.json > /dev/null
cp $(find $O/kt_bench -name '*kernel_stats.csv' | head -1) $O/kernel_stats_bench.csv 2>/dev/null
cp $(find $O/kt_train -name '*kernel_stats.csv' | head -1) $O/kernel_stats_train_step.csv 2>/dev/null
# the raw traces are large: keep the summaries only
rm -rf $O/kt_bench $O/kt_train $O/kt_train2 $O/pmc_traffic $O/pmc_traffic_bench $O/pmc_sq1 $O/pmc_sq2 $O/pmc_tree $O/pmc_board $O/pmc_board6 $O/pmc_board_t5 $O/pmc_board_t6
ls -la $O
step "done"
