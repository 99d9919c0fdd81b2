#!/bin/bash
# Round 3 (run on the GPU box through gpurun, from the repo root): the bench line with its `smoother` object, the kernel
# trace of the same command, and the lane smoother's HBM traffic (WRITE_SIZE / FETCH_SIZE, one rocprofv3 pass each).
# Summaries are copied to profiles/ by hand (pmc_summary.py for the counters).
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/prof_r03
mkdir -p $OUT
python3 bench.py > $OUT/bench_1.json 2>$OUT/bench_1.err; cut -c1-600 $OUT/bench_1.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_prof_line.json 2>$OUT/bench_prof.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_sm_w -- python3 tools/configs_r02.py --only 3s --reps 2 > /dev/null 2>$OUT/pmc_sm_w.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_sm_f -- python3 tools/configs_r02.py --only 3s --reps 2 > /dev/null 2>$OUT/pmc_sm_f.err
python3 tools/pmc_summary.py --kernel rts_smooth_lane_kernel --note "lane smoother (hand-managed AGPR file), 65 536 x 1 023 steps; WRITE_SIZE and FETCH_SIZE in KiB per launch, one rocprofv3 pass each" $OUT/pmc_sm_w $OUT/pmc_sm_f > $OUT/smoother_pmc.json
cat $OUT/smoother_pmc.json
find $OUT -name "*kernel_stats.csv" | head -3
f=$(find $OUT/bench -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" | cut -c1-200
echo profile_done
