#!/bin/bash
# The evidence run of a round on the MI355X box (`gpurun --timeout 3000 -- 'bash tools/gpu_evidence.sh r06'`): the whole GPU test
# suite, the three bench lines, rocprofv3 kernel traces of configs[1] / configs[2] reduced to step summaries, the kernel
# micro-benchmarks, the matrix-pipe PMC pass (counters in their own runs, --kernel-trace only) and the through-the-loop
# probe.  Everything lands in gpurun_out/<tag>/ ; copy what is to be judged into profiles/ with the <tag>_ prefix.
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
O=$R/gpurun_out/$TAG
rm -rf "$O"; mkdir -p "$O"
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -3 | tee "$O/tests.txt"
cp gpurun_out/parity_record.json "$O/parity.json"
for c in 1 2 3; do
  python bench.py --config $c 2>/dev/null | grep '^{' | tail -1 > "$O/bench_c$c.json"
  python -c "import json;d=json.load(open('$O/bench_c$c.json'));print('config $c', d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('through_loop',{}).get('value'))"
done
# kernel traces.  (a) ONE batch in flight: every kernel has the chip to itself - the per-kernel durations the rooflines are priced
# with, the launch-order listing of one iteration; (b) the default schedule (two batches in flight): rocprofv3's own per-kernel
# statistics of the run whose timed region is the bench line's (kernels of the two streams overlap: busy time > wall time)
for c in 1 2; do
  rm -rf /tmp/prof$c
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof$c -o r -- python "$R/bench.py" --config $c --steps 2 --warmup 1 --in-flight 1 --no-cpu-baseline --no-extras > "$O/prof_c$c.log" 2>&1)
  KT=$(find /tmp/prof$c -name '*kernel_trace.csv' | head -1); KS=$(find /tmp/prof$c -name '*kernel_stats.csv' | head -1)
  cp "$KS" "$O/bench_c${c}_kernel_stats.csv"
  python tools/trace_summary.py "$KT" "$O/bench_c${c}_step_summary.json" 40 > "$O/bench_c${c}_step_summary.txt" 2>&1
  python tools/trace_iteration.py "$KT" $([ $c = 1 ] && echo PgdLinfOp || echo pgd_l2_fused_kernel) > "$O/iteration_c$c.txt" 2>&1
  rm -rf /tmp/prof${c}b
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof${c}b -o r -- python "$R/bench.py" --config $c --steps 4 --warmup 2 --no-cpu-baseline --no-extras > "$O/prof_c${c}_two_in_flight.log" 2>&1)
  KS=$(find /tmp/prof${c}b -name '*kernel_stats.csv' | head -1); KT=$(find /tmp/prof${c}b -name '*kernel_trace.csv' | head -1)
  cp "$KS" "$O/bench_c${c}_two_in_flight_kernel_stats.csv"
  python tools/trace_summary.py "$KT" "$O/bench_c${c}_two_in_flight_step_summary.json" 40 > "$O/bench_c${c}_two_in_flight_step_summary.txt" 2>&1
done
python tools/two_stream_probe.py --config 1 --lanes 1,2,3 --rounds 2 2>&1 | grep -v amdgpu.ids > "$O/batches_in_flight_c1.txt"
python tools/two_stream_probe.py --config 2 --lanes 1,2,3 --rounds 2 2>&1 | grep -v amdgpu.ids > "$O/batches_in_flight_c2.txt"
python tools/coresidency_probe.py 2>&1 | grep -v amdgpu.ids > "$O/coresidency_probe.txt"
python tools/kernel_microbench.py --json "$O/kernel_microbench.json" 2>&1 | grep -v amdgpu.ids > "$O/kernel_microbench.txt"
python tools/model_kernel_bench.py --json "$O/model_kernel_microbench.json" 2>&1 | grep -v amdgpu.ids > "$O/model_kernel_microbench.txt"
python tools/specrnet_conv_probe.py 128 2>&1 | grep -v amdgpu.ids > "$O/specrnet_conv_probe.txt"
python tools/conv0_bwd_probe.py 2>&1 | grep -v amdgpu.ids > "$O/conv0_bwd_probe.txt"
rm -rf /tmp/pmc1 /tmp/pmc2
(cd /tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA --kernel-trace --output-format csv -d /tmp/pmc1 -- python "$R/tools/model_kernel_bench.py" --launches 3 > /dev/null 2>&1)
(cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d /tmp/pmc2 -- python "$R/tools/model_kernel_bench.py" --launches 3 > /dev/null 2>&1)
python tools/pmc_summary.py $(find /tmp/pmc1 /tmp/pmc2 -name '*counter_collection.csv') > "$O/model_kernel_pmc.txt" 2>&1
python tools/config_probe.py --batches 6 2>&1 | grep -v amdgpu.ids > "$O/config_probe.txt"
python -m tests.parity_attribution --out ${TAG}_parity_attribution > /dev/null 2>&1; cp gpurun_out/${TAG}_parity_attribution.* "$O/" 2>/dev/null
ls -la "$O"
