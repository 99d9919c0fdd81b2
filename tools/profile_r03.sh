#!/bin/bash
# Round 3 (run on the GPU box through gpurun, from the repo root): the bench line with its `smoother` object (three samples), the
# kernel trace of the same command, every BASELINE configuration, the lane smoother's HBM traffic (WRITE_SIZE / FETCH_SIZE, one
# rocprofv3 pass each), the D = 168 smoother's kernel trace and HBM traffic, and the distance of the D = 84 / 112 / 168 kernels
# from the extended-precision fixtures.  Summaries are copied to profiles/ by hand (pmc_summary.py for the counters).
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/prof_r03
mkdir -p $OUT
for k in 1 2 3; do python3 bench.py > $OUT/bench_$k.json 2>$OUT/bench_$k.err; cut -c1-200 $OUT/bench_$k.json; done
python3 tools/configs_r02.py --only 2,3x8,5,5x8,3s,4 > $OUT/configs.jsonl 2>$OUT/configs.err; cut -c1-200 $OUT/configs.jsonl
python3 tools/exact_ratios.py > $OUT/exact_ratios.jsonl 2>$OUT/exact_ratios.err; cut -c1-300 $OUT/exact_ratios.jsonl
python3 tools/bench_modes.py --traj 2048 --nsteps 64 --modes pleiades_smooth > $OUT/pleiades_smooth.json 2>$OUT/pleiades_smooth.err; cat $OUT/pleiades_smooth.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_prof_line.json 2>$OUT/bench_prof.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_sm_w -- python3 tools/configs_r02.py --only 3s --reps 2 > /dev/null 2>$OUT/pmc_sm_w.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_sm_f -- python3 tools/configs_r02.py --only 3s --reps 2 > /dev/null 2>$OUT/pmc_sm_f.err
python3 tools/pmc_summary.py --kernel rts_smooth_lane_kernel --note "lane smoother (hand-managed AGPR file, L D L'), 65 536 x 1 023 steps; WRITE_SIZE and FETCH_SIZE in KiB per launch, one rocprofv3 pass each (FETCH_SIZE x 2 per the guide)" $OUT/pmc_sm_w $OUT/pmc_sm_f > $OUT/smoother_pmc.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pl_sm -- python3 tools/bench_modes.py --traj 2048 --nsteps 64 --modes pleiades_smooth > /dev/null 2>$OUT/pl_sm.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_pl_w -- python3 tools/bench_modes.py --traj 2048 --nsteps 64 --modes pleiades_smooth > /dev/null 2>$OUT/pmc_pl_w.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_pl_f -- python3 tools/bench_modes.py --traj 2048 --nsteps 64 --modes pleiades_smooth > /dev/null 2>$OUT/pmc_pl_f.err
python3 tools/pmc_summary.py --kernel rts_smooth_mfma_kernel --note "D = 168 smoother, 2 048 x 64, split pass: the workspace kernel (finish record r + 1 | begin record r), KiB per launch (126 launches per pass); FETCH_SIZE x 2 per the guide" $OUT/pmc_pl_w $OUT/pmc_pl_f > $OUT/pleiades_smoother_mfma_pmc.json
python3 tools/pmc_summary.py --kernel rts_smooth_sweeps_kernel --note "D = 168 smoother, 2 048 x 64, split pass: factorisation and sweeps on chip, KiB per launch (63 launches per pass)" $OUT/pmc_pl_w $OUT/pmc_pl_f > $OUT/pleiades_smoother_sweeps_pmc.json
cat $OUT/smoother_pmc.json $OUT/pleiades_smoother_mfma_pmc.json $OUT/pleiades_smoother_sweeps_pmc.json
for f in $(find $OUT/bench $OUT/pl_sm -name "*kernel_stats.csv"); do echo $f; head -6 "$f" | cut -c1-180; done
echo profile_done
