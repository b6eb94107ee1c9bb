#!/bin/bash
# Round 6 measurement set on the GPU box (from the repo root):  profiles/tools/record_round6.sh <tag> [precision]
#   bench JSON lines (bench shape with every extra leg; C2), rocprofv3 kernel stats of the bench command, the FETCH_SIZE / WRITE_SIZE counter
#   passes and one SQ / GRBM counter pass (separate runs, kernel-trace only: gpurun refuses --pmc together with the other trace domains).
# The summaries bench.py attaches to its line are written as profiles/round6_tgt_<precision>_{pmc_traffic,sq_counters}.json by the caller
# (copied from gpurun_out/<tag>/).
set -o pipefail
tag=${1:-x}
prec=${2:-fp16}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python3 bench.py --precision $prec --steps 10 --warmup 3 > $out/tgt_bench.json 2> $out/tgt_bench.log || exit 1
python3 bench.py --precision $prec --workload c2 --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $out/c2_bench.json 2> $out/c2_bench.log || exit 1
P="--precision $prec --no-cpu-baseline --no-sampling --no-extras --no-graph --steps 3 --warmup 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o tgt -- python3 bench.py $P > $out/stats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/pmc_fetch --output-format csv -- python3 bench.py $P > $out/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/pmc_write --output-format csv -- python3 bench.py $P > $out/pmc_write.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
    -d $out/pmc_sq --output-format csv -- python3 bench.py $P > $out/pmc_sq.log 2>&1 || exit 1
label="TGT [1024,256,88,5] $prec, eager launches, 3 steps + 1 warm-up"
sha=$(python3 profiles/tools/source_hash.py)          # the build these counters belong to: bench.py only lets them stand beside live timings of the same sources
python3 profiles/pmc_traffic.py $out/pmc_fetch $out/pmc_write $out/pmc_traffic.json "$label" 4 $sha > $out/pmc_traffic.txt
python3 profiles/sq_counters.py $out/pmc_sq $out/sq_counters.json "$label" $sha > $out/sq_counters.txt
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
find $out -name "*.db" -delete; find $out -name "*counter_collection.csv" -delete; find $out -name "*kernel_trace.csv" -size +5M -delete
ls -la $out
# side configurations and the f32 leg: bench lines of C3 / C4, the two sampling scans, kernel stats of the fp32 mode
python3 bench.py --workload c3 > $out/c3_bench.json 2> $out/c3_bench.log
MULTINN_JAMMING_GROUP=0 python3 bench.py --workload c3 > $out/c3_sequential_bench.json 2> $out/c3_sequential_bench.log     # one generator after the other (round 5's form)
python3 bench.py --workload c4 > $out/c4_bench.json 2> $out/c4_bench.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_c3 -o c3 -- python3 bench.py --workload c3 --no-graph --steps 3 --warmup 1 > $out/stats_c3.log 2>&1
cp $(find $out/stats_c3 -name "*kernel_stats.csv" | head -1) $out/c3_kernel_stats.csv
python3 profiles/tools/bench_generate.py > $out/generate_scan.json 2>&1
python3 profiles/tools/bench_feedback.py > $out/feedback_scan.json 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats32 -o fp32 -- python3 bench.py --precision fp32 --no-cpu-baseline --no-sampling --no-extras --no-graph --steps 2 --warmup 1 > $out/stats32.log 2>&1
cp $(find $out/stats32 -name "*kernel_stats.csv" | head -1) $out/fp32_kernel_stats.csv
find $out -name "*.db" -delete; find $out -name "*kernel_trace.csv" -size +5M -delete
ls -la $out
