#!/bin/bash
# One measurement pass on the GPU box (profiles/README.md): bench line, kernel stats, PMC traffic of the attention kernel, MFMA-busy.
# Run as: gpurun -- bash tools/measure_pass.sh   (outputs under gpurun_out/pass; copy what is to be judged into profiles/)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pass
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p6 /tmp/pm /tmp/mf /tmp/pr     # (a box can be handed out again with its /tmp: never read an earlier run's trace)
timeout 400 python3 $R/bench.py > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.json
timeout 200 python3 $R/bench.py --no-f32 --no-cpu-baseline --long-run 0 --rows-per-gpu-probe 80 2>/dev/null | python3 -c "import sys, json; print(json.dumps(json.loads(sys.stdin.readline())['strong_scaling_probe']))" > $O/strong_scaling_probe.json
cat $O/strong_scaling_probe.json
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p6 -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32 --long-run 0 > /tmp/p6.log 2>&1
python3 $R/tools/prof_summary.py /tmp/p6 32 45 > $O/bench_summary.txt 2>&1
python3 $R/tools/step_timeline.py $(find /tmp/p6 -name "*kernel_trace.csv" | head -1) 4 --full > $O/step_timeline.txt 2>&1
grep '"metric"' /tmp/p6.log | tail -1 > $O/bench_under_rocprof.json   # the same process's own HIP-event figures
cp $(find /tmp/p6 -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pm/fetch -- python3 $R/tools/attn_kernel_only.py > /tmp/pmf.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pm/write -- python3 $R/tools/attn_kernel_only.py > /tmp/pmw.log 2>&1
python3 $R/tools/pmc_traffic.py /tmp/pm attn_fwd_fast_kernel $O/attn_fwd_pmc_bf16.json
cp $(find /tmp/pm/fetch -name "*counter_collection.csv" | head -1) $O/attn_fwd_pmc_fetch.csv
cp $(find /tmp/pm/write -name "*counter_collection.csv" | head -1) $O/attn_fwd_pmc_write.csv
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/mf -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-f32 --long-run 0 > /tmp/mf.log 2>&1
python3 $R/tools/pmc_mfma.py /tmp/mf 12 > $O/mfma_busy.txt 2>&1
head -12 $O/mfma_busy.txt
tail -5 $O/bench_summary.txt
cat $O/attn_fwd_pmc_bf16.json
# where the fused step's time goes without a profiler attached (timing marks recorded by the step itself) and the host time of its C call
timeout 200 python3 $R/tools/host_time.py > $O/step_marks.txt 2>&1
tail -18 $O/step_marks.txt
# the same with UIC_REC_EARLY_GRADS: when each gradient group is final, bytes final in the step's last 0.1 ms
timeout 200 python3 $R/tools/host_time.py --early > $O/step_marks_early_grads.txt 2>&1
tail -7 $O/step_marks_early_grads.txt
# persistent recurrence kernel: phase stamps, parity against the per-step launch chain, forward time in modes 0 / 1 / 2
timeout 200 python3 $R/tools/rnn_persist_probe.py --dbg > $O/rnn_persist_probe.txt 2>&1
grep -E "phase|step  |forward" $O/rnn_persist_probe.txt
timeout 200 python3 $R/tools/gemm_headroom.py > $O/gemm_headroom.txt 2>&1
# persistent BPTT kernel (opt-in): parity with the launch chain, phase stamps, backward call and fused step in both modes
timeout 300 python3 $R/tools/rnn_bwd_probe.py --dbg > $O/rnn_bwd_probe.txt 2>&1
grep -E "step |fused|backward call" $O/rnn_bwd_probe.txt
# persistent decode launch against the per-step launch chain: agreement, pass times, phase stamps
timeout 400 python3 $R/tools/decode_probe.py --dbg 2>/dev/null > $O/decode_probe.txt
cat $O/decode_probe.txt
# attention-step kernel variants from a C++ host (shipped structure, loads only, online softmax, wave counts)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $R/tools/micro/attn_variants.hip -o /tmp/attn_variants 2>/dev/null && /tmp/attn_variants > $O/attn_variants.txt 2>&1
# HBM traffic of the persistent recurrence kernel (separate PMC passes, same corrections as for the attention kernel)
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pr/fetch -- python3 $R/tools/rnn_kernel_only.py > /tmp/prf.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pr/write -- python3 $R/tools/rnn_kernel_only.py > /tmp/prw.log 2>&1
tail -2 /tmp/prf.log
python3 $R/tools/pmc_traffic.py /tmp/pr rnn_fwd_persist $O/rnn_persist_pmc_bf16.json
# secondary paths on the same box: API / decode / self-critical phases, pivot NMT training step, joint pivot decode
timeout 300 python3 $R/tools/phase_times.py > $O/phase_times.txt 2>/dev/null
timeout 200 python3 $R/tools/scst_phases.py 2>/dev/null | tail -21 > $O/scst_phases.txt
timeout 200 python3 $R/tools/nmt_bench.py --steps 20 2>/dev/null | tail -1 > $O/nmt_bench.txt
timeout 200 python3 $R/tools/pivot_decode_bench.py --iters 20 2>/dev/null | tail -1 > $O/pivot_decode.txt
timeout 300 bash $R/tools/nmt_profile.sh > $O/nmt_profile.txt 2>&1
timeout 300 bash $R/tools/pivot_profile.sh > $O/pivot_profile.txt 2>&1
cat $O/nmt_bench.txt $O/pivot_decode.txt
# round 5: the weight-gradient kernels per split-K, the stream-count probe, the overlapped exchange with the one-GPU stand-in
timeout 300 python3 $R/tools/tn_bench.py > $O/tn_bench.txt 2>&1
timeout 200 python3 $R/tools/queue_probe.py 0 > $O/queue_probe.txt 2>&1
echo "--- GPU_MAX_HW_QUEUES=2" >> $O/queue_probe.txt
GPU_MAX_HW_QUEUES=2 timeout 200 python3 $R/tools/queue_probe.py 0 >> $O/queue_probe.txt 2>&1
# round 6: the sharded exchange (reduce-scatter / Adam on the shard / all-gather) and round 5's all-reduce beside the step, stand-ins
# for RCCL on one GPU; with the two hardware queues a data-parallel Trainer insists on, and without
GPU_MAX_HW_QUEUES=2 timeout 300 python3 $R/tools/comm_proxy.py --steps 20 > $O/comm_proxy.txt 2>&1
echo "--- opt.bf16_gradient_exchange (the pieces behind the join reduce-scattered as bf16)" >> $O/comm_proxy.txt
GPU_MAX_HW_QUEUES=2 timeout 300 python3 $R/tools/comm_proxy.py --steps 20 --only sharded --half >> $O/comm_proxy.txt 2>&1
echo "--- without GPU_MAX_HW_QUEUES" >> $O/comm_proxy.txt
timeout 300 python3 $R/tools/comm_proxy.py --steps 20 >> $O/comm_proxy.txt 2>&1
grep -E "^no exchange|^SHARDED|^round 5|^---" $O/comm_proxy.txt
# round 6: the file-backed loader (stored and deflated members) against resident batches, and training from files
timeout 300 python3 $R/tools/loader_bench.py > $O/loader_bench.txt 2>&1
echo "--- --compressed" >> $O/loader_bench.txt
timeout 300 python3 $R/tools/loader_bench.py --compressed >> $O/loader_bench.txt 2>&1
tail -12 $O/loader_bench.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $R/tools/micro/mfma_agpr.hip -o /tmp/mfma_agpr 2>/dev/null && timeout 100 /tmp/mfma_agpr > $O/mfma_agpr.txt 2>&1
timeout 200 python3 $R/tools/ab_knobs.py 0 0x200 0x600 > $O/ab_knobs.txt 2>&1
cat $O/ab_knobs.txt
# round 5 (second half): per-phase stamps of the pivot decoder's persistent launches; what a dependent launch costs
timeout 300 python3 $R/tools/nmt_bwd_probe.py > $O/nmt_bwd_probe.txt 2>&1
tail -24 $O/nmt_bwd_probe.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $R/tools/micro/chain_probe.hip -o /tmp/chain_probe 2>/dev/null && timeout 200 /tmp/chain_probe > $O/chain_probe.txt 2>&1
timeout 400 python3 $R/tools/scst_bench.py > $O/scst_bench.txt 2>&1
tail -5 $O/scst_bench.txt
