#!/bin/bash
# Round-2 profiles (run on the GPU box through gpurun, from the repo root): rocprofv3 kernel traces of the headline bench and
# of BASELINE configs 2 / 5 / the 8-GPU shard of config 3 with the kernels the launcher picks, then separate PMC passes
# (WRITE_SIZE, FETCH_SIZE) for the row-team filter + smoother of config 2.  Summaries are copied to profiles/ by hand.
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/prof_r02
mkdir -p $OUT
python3 tools/configs_r02.py --only 2,3x8,5,5x8,3s,4 > $OUT/configs.jsonl 2>$OUT/configs.err && cat $OUT/configs.jsonl
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_line.json 2>$OUT/bench.err && cat $OUT/bench_line.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg2 -- python3 tools/configs_r02.py --only 2 --reps 3 > /dev/null 2>$OUT/cfg2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg5 -- python3 tools/configs_r02.py --only 5 --reps 3 > /dev/null 2>$OUT/cfg5.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg3x8 -- python3 tools/configs_r02.py --only 3x8 --reps 3 > /dev/null 2>$OUT/cfg3x8.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_w -- python3 tools/configs_r02.py --only 2 --reps 2 > /dev/null 2>$OUT/pmc_w.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_f -- python3 tools/configs_r02.py --only 2 --reps 2 > /dev/null 2>$OUT/pmc_f.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq -- python3 tools/configs_r02.py --only 2 --reps 2 > /dev/null 2>$OUT/pmc_sq.err
find $OUT -name "*kernel_stats.csv" | head; echo profile_done
