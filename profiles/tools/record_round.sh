#!/bin/bash
# Records one measurement set on the GPU box: bench JSON lines (bench shape and C2), rocprofv3 kernel stats of the bench command, and the
# FETCH_SIZE / WRITE_SIZE counter passes (separate runs, kernel-trace only).  Usage: profiles/tools/record_round.sh <tag>   (from the repo root)
set -o pipefail
tag=${1:-x}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python3 bench.py --steps 10 --warmup 3 > $out/tgt_bench.json 2> $out/tgt_bench.log || exit 1
python3 bench.py --workload c2 --steps 20 --warmup 3 --no-cpu-baseline > $out/c2_bench.json 2> $out/c2_bench.log || exit 1
P="--no-cpu-baseline --no-sampling --no-extras --no-graph --steps 3 --warmup 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o tgt -- python3 bench.py $P > $out/stats.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/pmc_fetch --output-format csv -- python3 bench.py $P > $out/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/pmc_write --output-format csv -- python3 bench.py $P > $out/pmc_write.log 2>&1 || exit 1
python3 profiles/pmc_traffic.py $out/pmc_fetch $out/pmc_write $out/pmc_traffic.json "TGT [1024,256,88,5] bf16, eager launches, 3 steps + 1 warm-up" > $out/pmc_traffic.txt
find $out -name "*.db" -delete; find $out -name "*counter_collection.csv" -size +20M -delete
ls -la $out
