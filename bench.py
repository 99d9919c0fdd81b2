#!/usr/bin/env python3
"""Headline benchmark: ensemble ODE-filter steps/sec on the BASELINE.json workload.

One "step" of this script = one pass of the hot path (fixed-step EK1(order=3) filter with
every step saved, as the reference's `solve` does: src/integrator_utils.jl:33-48) over one
ensemble of `--traj` Lorenz-63 trajectories x `--nsteps` solver steps per GPU.  Inputs are
generated on the device (splitmix64 ensemble) before the timed region; output
buffers are allocated once and stay resident in HBM.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

STRONG scaling is the default and the headline (`value`): BASELINE.json's 65 536-trajectory ensemble is split evenly
over the ranks (contiguous blocks, global trajectory numbering, so every rank count solves the SAME ensemble); no
exchange while stepping; one RCCL all-gather of the final posterior means at the end of each pass, inside the timed
region.  With more than one rank the line also carries a `weak_scaling` object: the same loop with 65 536
trajectories PER GPU (`--mode weak` makes that the headline instead).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_ALG_STEP = lambda D: 8 * (D + D * (D + 1) // 2 + 1)  # bytes written per trajectory-step
HBM_PEAK = 8.0e12  # B/s, MI355X_MICROARCH.md (spec; 6.29e12 measured streaming copy)


def cpu_baseline(seconds_budget=15.0):
    """The oracle (numpy restatement of the reference arithmetic, kind='port') or, when built,
    its C twin, timed on this box's host cores on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    try:
        import subprocess

        # rebuild for THIS host's cores (-march=native); the checker's build, not the product's
        subprocess.run(["make", "-B", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        import odefilter_cport as cport  # C restatement of the reference loop, OpenMP over trajectories

        if cport.available():
            return cport.bench_lorenz(seconds_budget)
    except Exception as e:  # fall back to the numpy oracle (slower, 1 thread)
        print(f"[bench] C restatement unavailable ({e}); timing the numpy oracle instead", file=sys.stderr)
    import odefilter_oracle as orc

    vf = orc.vector_field("lorenz63")
    u0s = orc.ensemble_u0(vf.u0, 64, 1e-2)
    nsteps = 256
    done, t0 = 0, time.perf_counter()
    for i in range(64):
        orc.solve(vf, orc.EK1(order=3, smooth=False), u0=u0s[i], tspan=(0.0, nsteps * 2.0**-9), dt=2.0**-9)
        done += nsteps
        if time.perf_counter() - t0 > seconds_budget:
            break
    el = time.perf_counter() - t0
    return {"value": done / el, "unit": "filter steps/s", "cores": 1, "kind": "port",
            "sample": f"{done // nsteps} trajectories x {nsteps} steps of the same Lorenz-63 EK1(3) workload, numpy oracle, 1 thread"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)  # the first launches after allocation run 5-8 % slower (first touch of 48.9 GB, clock ramp)
    ap.add_argument("--mode", choices=["strong", "weak"], default="strong",
                    help="strong (default): --total-traj split over the ranks; weak: --traj per GPU")
    ap.add_argument("--total-traj", type=int, default=65536, help="ensemble size of the strong-scaling run (BASELINE config 3)")
    ap.add_argument("--traj", type=int, default=65536, help="trajectories per GPU of the weak-scaling run")
    ap.add_argument("--nsteps", type=int, default=1024, help="solver steps per trajectory (tspan = nsteps * 2^-9)")
    ap.add_argument("--save", choices=["everystep", "final"], default="everystep")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-smoother", action="store_true", help="skip the RTS smoother pass measured after the timed loop")
    args = ap.parse_args()

    import torch

    import odefilters_jl_amd as pkg
    from odefilters_jl_amd import dist as od

    rank, world, local = od.init_from_env(backend="nccl")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    d, q = 3, 3
    D = d * (q + 1)
    TRI = D * (D + 1) // 2
    nsteps = args.nsteps
    everystep = args.save == "everystep"
    n_save = nsteps + 1 if everystep else 1
    dt = 2.0**-9
    tgrid = np.arange(nsteps + 1) * dt
    stream = torch.cuda.current_stream(dev)

    def run(mode, steps, warmup, with_smoother=False):
        """One timed loop.  Returns (seconds for `steps` passes [max over ranks], kernel ms, N per rank, parity info, ...).
        with_smoother: after the timed loop (outside it) the RTS smoother runs over the records of the last pass."""
        if mode == "strong":
            lo, hi = od.shard_bounds(args.total_traj, rank, world)
            N, first = hi - lo, lo
        else:
            N, first = args.traj, rank * args.traj
        # (smooth = True only sizes the context for the RTS pass measured AFTER the timed loop; solve_fixed runs the filter alone)
        ctx = pkg.Context("lorenz63", q, 1, N, save_everystep=everystep, smooth=everystep and with_smoother, device=local, want_loglik=True)
        ctx.set_stream(stream.cuda_stream)
        # resident output buffers owned by torch (so the all-gather reads them in place)
        mean = torch.empty((n_save, D, N), dtype=torch.float64, device=dev)
        cov = torch.empty((n_save, TRI, N), dtype=torch.float64, device=dev)
        ctx.bind_device(0, mean.data_ptr(), mean.numel() * 8)
        ctx.bind_device(1, cov.data_ptr(), cov.numel() * 8)
        smean = scov = None
        if with_smoother and everystep:  # the smoothed records, likewise resident (another 48.9 GB of the 288)
            from odefilters_jl_amd.host import F_SMOOTH_MEAN, F_SMOOTH_COV_TRIL
            smean = torch.empty((n_save, D, N), dtype=torch.float64, device=dev)
            scov = torch.empty((n_save, TRI, N), dtype=torch.float64, device=dev)
            ctx.bind_device(F_SMOOTH_MEAN, smean.data_ptr(), smean.numel() * 8)
            ctx.bind_device(F_SMOOTH_COV_TRIL, scov.data_ptr(), scov.numel() * 8)
        ctx.set_problem_perturbed([1.0, 0.0, 0.0], [10.0, 28.0, 8.0 / 3.0], 0.0, 1e-2, first_index=first)
        n_pad = -(-max(args.total_traj, 1) // world) if mode == "strong" else N  # equal blocks for the all-gather
        send = torch.zeros((D, n_pad), dtype=torch.float64, device=dev) if world > 1 else None
        kernel_ms = []

        def one_pass():
            ctx.solve_fixed(tgrid)  # launches on torch's stream; records hipEvents around the kernel
            kernel_ms.append(ctx.kernel_time_ms(0)[0])
            if world > 1:
                send[:, :N].copy_(mean[n_save - 1])
                return od.allgather_shards(send, world)
            return mean[n_save - 1]

        def sync_all():
            torch.cuda.synchronize(dev)
            if world > 1:
                torch.distributed.barrier()
            torch.cuda.synchronize(dev)

        for _ in range(warmup):
            one_pass()
        kernel_ms.clear()
        sync_all()
        t0 = time.perf_counter()
        for _ in range(steps):
            final = one_pass()
        sync_all()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            el = float(t.item())
        finite_ok = bool(torch.isfinite(final).all().item()) and bool((ctx.get(10) == 0).all())
        # parity inside the bench: THIS run's records against the committed oracle fixture (tests/golden/full_lorenz_fixed.npz: the
        # same ensemble, 1 024 steps; seven trajectories at seven record times 0 ... 1 024) -- solution means of the seven records
        # at 1e-10 relative, derivative blocks and covariances at the full-size tests' bars (tests/test_gpu_fullsize.py)
        fx_path = os.path.join(ROOT, "tests", "golden", "full_lorenz_fixed.npz")
        fx = np.load(fx_path) if (rank == 0 and os.path.exists(fx_path) and nsteps == 1024) else None

        def unpack(tril):  # [TRI] -> [D, D]
            m = np.zeros((D, D))
            m[np.tril_indices(D)] = tril
            return m + np.tril(m, -1).T

        def against_fixture(mean_t, cov_t, key_m, key_c):
            worst_u, worst_all, worst_cov, n_cmp = 0.0, 0.0, 0.0, 0
            steps_fx = [int(v) for v in fx["steps"]] if everystep else [nsteps]
            for k, gi in enumerate(fx["idx"]):
                if not (first <= gi < first + N):
                    continue
                n_cmp += 1
                for j, st in enumerate(fx["steps"]):
                    if int(st) not in steps_fx:
                        continue
                    slot = int(st) if everystep else 0
                    m = mean_t[slot, :, gi - first].cpu().numpy()
                    ref = fx[key_m][k][j]
                    worst_u = max(worst_u, float(np.abs(m[:d] - ref[:d]).max() / np.abs(ref[:d]).max()))
                    for b in range(1, q + 1):  # derivative blocks, relative to the block's size
                        sl = slice(b * d, (b + 1) * d)
                        worst_all = max(worst_all, float(np.abs(m[sl] - ref[sl]).max() / (np.abs(ref[sl]).max() + 1e-300)))
                    c = unpack(cov_t[slot, :, gi - first].cpu().numpy())
                    rc_ = fx[key_c][k][j]
                    worst_cov = max(worst_cov, float(np.abs(c - rc_).max() / (np.abs(rc_).max() + 1e-300)))
            return {"max_rel_err_vs_oracle_fixture": worst_u, "max_rel_err_derivative_blocks": worst_all, "max_rel_err_covariance": worst_cov,
                    "trajectories_compared": n_cmp, "records_compared_per_trajectory": len(steps_fx), "tolerance": 1e-10,
                    "tolerance_derivative_blocks": 1e-6, "tolerance_covariance": 5e-3,
                    "ok": bool(n_cmp >= 4 and worst_u <= 1e-10 and worst_all <= 1e-6 and worst_cov <= 5e-3)}

        parity = against_fixture(mean, cov, "mean_filt", "cov_filt") if fx is not None else None
        k_ms = float(np.mean(kernel_ms))
        kname = ctx.kernel_name(0)  # the kernel the library's launcher picked for this ensemble size (odef_kernel_name)
        smoother = None
        if with_smoother and everystep:
            # The reference's default is smooth = true (src/algorithms.jl:46-51): the RTS pass over the records of the last filter
            # pass, measured here, OUTSIDE the timed region (the headline metric is filter steps/s).
            s_ms = []
            for _ in range(4):
                ctx.smooth()
                s_ms.append(ctx.kernel_time_ms(1)[0])
            sm_ms = float(np.median(s_ms[1:]))
            sb = (2 * B_ALG_STEP(D) - 8) * N * (nsteps - 1)  # read the filter record, write the smoothed one
            smoother = {"kernel": ctx.kernel_name(1), "kernel_ms": sm_ms, "steps_per_s": N * (nsteps - 1) / (sm_ms * 1e-3),
                        "roofline": {"bound": "hbm", "achieved": sb / (sm_ms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                                     "frac": sb / (sm_ms * 1e-3) / HBM_PEAK, "algorithmic_bytes_per_launch": sb},
                        "filter_plus_smoother_ms": k_ms + sm_ms, "retcodes_ok": bool((ctx.get(10) == 0).all())}
            if fx is not None:
                smoother["parity"] = against_fixture(smean, scov, "mean_smooth", "cov_smooth")
        ctx.close()
        del mean, cov, smean, scov
        torch.cuda.empty_cache()
        return el, k_ms, N, finite_ok, parity, kname, smoother

    el, k_ms, N, finite_ok, parity, kname, smoother = run(args.mode, args.steps, args.warmup, with_smoother=not args.no_smoother)
    total_traj = args.total_traj if args.mode == "strong" else world * N
    value = total_traj * nsteps * args.steps / el
    alg_bytes = B_ALG_STEP(D) * N * (nsteps + 1 if everystep else 1)
    achieved = alg_bytes / (k_ms * 1e-3)
    traffic, traffic_source = None, None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            key = f"lorenz63_ek1q3_N{N}_n{nsteps}_{args.save}"
            traffic = tj.get(key, {}).get("hbm_bytes_per_launch")
            if traffic is not None:  # NOT measured in this run: PMC counters need their own rocprofv3 passes
                traffic_source = "profiles/hbm_traffic.json (separate rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE passes of the same launch)"
        except Exception:
            traffic = None
    other = None
    if world > 1:  # the other scaling mode beside the headline
        om = "weak" if args.mode == "strong" else "strong"
        el2, k2, N2, _, _, kname2, _ = run(om, max(2, args.steps // 2), 1)
        tot2 = args.total_traj if om == "strong" else world * N2
        other = {"scaling": om, "value": tot2 * nsteps * max(2, args.steps // 2) / el2, "unit": "filter steps/s",
                 "trajectories_per_gpu": N2, "ms_per_step": el2 / max(2, args.steps // 2) * 1e3, "kernel_ms": k2, "kernel": kname2}

    if rank == 0:
        line = {
            "metric": "ensemble ODE-filter steps/sec (whole node)",
            "value": value,
            "unit": "filter steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": el / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.mode,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"Lorenz-63 d=3, EK1(order=3), dynamic diffusion, fixed dt=2^-9, {total_traj} trajectories in total = {N} per GPU "
                            f"x {nsteps} steps, save={args.save} (BASELINE.json configs[2]; perturbed u0 via splitmix64)",
                "total_trajectories": total_traj, "trajectories_per_gpu": N, "solver_steps": nsteps, "state_dim": D,
                "parallelism": f"ensemble-shard x{world}",
                "collective": "one all_gather of final means per pass" if world > 1 else "none",
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": achieved / HBM_PEAK, "traffic": traffic, "traffic_source": traffic_source,
                "kernel": kname,
                "kernel_ms": k_ms,
                "algorithmic_bytes_per_launch": alg_bytes,
            },
            "finite_ok": finite_ok,
            "parity_ok": bool(parity["ok"]) if parity else None,
            "parity": parity,
        }
        if smoother:
            line["smoother"] = smoother
        if other:
            line[other["scaling"] + "_scaling"] = other
        if world > 1:
            # No multi-GPU node was available to any round of this build: a multi-rank run of this script is the FIRST execution of
            # the sharded path on more than one device (the CPU suite covers it with world-size-2 gloo, the GPU suite with one rank).
            line["first_multi_device_execution"] = True
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
            line["gpu_over_cpu"] = value / line["cpu_baseline"]["value"]
        print(json.dumps(line))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
