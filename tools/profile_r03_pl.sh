#!/bin/bash
# Round 3, D = 168 smoother only (run on the GPU box through gpurun, from the repo root): kernel trace and HBM traffic
# (WRITE_SIZE / FETCH_SIZE, one rocprofv3 pass each) of the split pass at 2 048 trajectories x 64 steps.
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/prof_r03_pl
mkdir -p $OUT
python3 tools/bench_modes.py --traj 2048 --nsteps 64 --modes pleiades_smooth > $OUT/pleiades_smooth.json 2>$OUT/pleiades_smooth.err; cat $OUT/pleiades_smooth.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pl_sm -- python3 tools/bench_modes.py --traj 2048 --nsteps 64 --modes pleiades_smooth > /dev/null 2>$OUT/pl_sm.err
if [ "$1" = "pmc" ]; then
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_pl_w -- python3 tools/bench_modes.py --traj 2048 --nsteps 64 --modes pleiades_smooth > /dev/null 2>$OUT/pmc_pl_w.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_pl_f -- python3 tools/bench_modes.py --traj 2048 --nsteps 64 --modes pleiades_smooth > /dev/null 2>$OUT/pmc_pl_f.err
python3 tools/pmc_summary.py --kernel rts_smooth_predict_kernel --note "D = 168 smoother, 2 048 x 64, split pass: unpack + predict (one thread per component pair, the packed record in LDS), KiB per launch; FETCH_SIZE x 2 per the guide" $OUT/pmc_pl_w $OUT/pmc_pl_f > $OUT/pleiades_smoother_predict_pmc.json
python3 tools/pmc_summary.py --kernel rts_smooth_sweeps_kernel --note "D = 168 smoother, 2 048 x 64, split pass: factorisation, sweeps, mean, G M G' and the smoothed record on chip, KiB per launch; FETCH_SIZE x 2 per the guide" $OUT/pmc_pl_w $OUT/pmc_pl_f > $OUT/pleiades_smoother_sweeps_pmc.json
cat $OUT/pleiades_smoother_predict_pmc.json $OUT/pleiades_smoother_sweeps_pmc.json
fi
for f in $(find $OUT/pl_sm -name "*kernel_stats.csv"); do echo $f; head -8 "$f" | cut -c1-200; done
echo profile_done
