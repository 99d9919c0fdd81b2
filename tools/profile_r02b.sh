#!/bin/bash
# Round-2, second half (run on the GPU box through gpurun, from the repo root): the numbers behind DESIGN.md sections 3.8-3.10 / 5
# for the final binary -- all configurations, three bench lines, kernel traces of the bench / config 4 / config 3 + smoother,
# PMC passes (each in a run of its own) for the lane smoother's traffic and the D = 168 MFMA filter's instruction mix, the two
# phase-stamp tools and the MFMA unit test.  Summaries are copied to profiles/ by hand.
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/prof_r02b
mkdir -p $OUT
python3 tools/configs_r02.py --only 2,3x8,5,5x8,3s,4 > $OUT/configs.jsonl 2>$OUT/configs.err && cat $OUT/configs.jsonl | cut -c1-260
ODEF_PLEIADES_FILTER=tiles python3 tools/configs_r02.py --only 4 > $OUT/config4_tiles.jsonl 2>>$OUT/configs.err
for k in 1 2 3; do python3 bench.py > $OUT/bench_$k.json 2>$OUT/bench_$k.err; cut -c1-400 $OUT/bench_$k.json; done
python3 tools/bench_modes.py --traj 2048 --nsteps 64 --modes pleiades_smooth > $OUT/pleiades_smooth.json 2>>$OUT/configs.err
# (round 2 only: the switch left the library in round 3) ODEF_PLEIADES_SMOOTH=team python3 tools/bench_modes.py --traj 2048 --nsteps 64 --modes pleiades_smooth >> $OUT/pleiades_smooth.json 2>>$OUT/configs.err
cat $OUT/pleiades_smooth.json | cut -c1-300
timeout -k 5 100 tools/mfma_filter_stamps > $OUT/mfma_filter_stamps.txt 2>&1
timeout -k 5 60 tools/mfma_dense_test > $OUT/mfma_dense_test.txt 2>&1; timeout -k 5 60 tools/mfma_layout_test > $OUT/mfma_layout_test.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_prof_line.json 2>$OUT/bench_prof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg4 -- python3 tools/configs_r02.py --only 4 --reps 4 > /dev/null 2>$OUT/cfg4.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg3s -- python3 tools/configs_r02.py --only 3s --reps 3 > /dev/null 2>$OUT/cfg3s.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_sm_w -- python3 tools/configs_r02.py --only 3s --reps 2 > /dev/null 2>$OUT/pmc_sm_w.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_sm_f -- python3 tools/configs_r02.py --only 3s --reps 2 > /dev/null 2>$OUT/pmc_sm_f.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT/pmc_p_sq -- python3 tools/configs_r02.py --only 4 --reps 2 > /dev/null 2>$OUT/pmc_p_sq.err
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_p_mfma -- python3 tools/configs_r02.py --only 4 --reps 2 > /dev/null 2>$OUT/pmc_p_mfma.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_p_f -- python3 tools/configs_r02.py --only 4 --reps 2 > /dev/null 2>$OUT/pmc_p_f.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_p_w -- python3 tools/configs_r02.py --only 4 --reps 2 > /dev/null 2>$OUT/pmc_p_w.err
find $OUT -name "*kernel_stats.csv" | head; tail -3 $OUT/pmc_p_mfma.err; echo profile_done
