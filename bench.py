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

Weak scaling: every rank (GPU) filters its own `--traj` trajectories (global trajectory index
= rank * traj + i); no exchange while stepping; one RCCL all-gather of the final posterior
means at the end of each pass (inside the timed region).
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
    ap.add_argument("--traj", type=int, default=65536, help="trajectories per GPU (weak scaling, the default)")
    ap.add_argument("--total-traj", type=int, default=0,
                    help="strong scaling instead: this many trajectories in total, split evenly over the ranks")
    ap.add_argument("--nsteps", type=int, default=1024, help="solver steps per trajectory (tspan = nsteps * 2^-9)")
    ap.add_argument("--save", choices=["everystep", "final"], default="everystep")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
    N, nsteps = args.traj, args.nsteps
    if args.total_traj:
        if args.total_traj % world:
            raise SystemExit("--total-traj must be divisible by the number of ranks")
        N = args.total_traj // world
    everystep = args.save == "everystep"
    n_save = nsteps + 1 if everystep else 1
    dt = 2.0**-9
    tgrid = np.arange(nsteps + 1) * dt

    ctx = pkg.Context("lorenz63", q, 1, N, save_everystep=everystep, smooth=False, device=local, want_loglik=True)
    stream = torch.cuda.current_stream(dev)
    ctx.set_stream(stream.cuda_stream)
    # resident output buffers owned by torch (so the all-gather reads them in place)
    mean = torch.empty((n_save, D, N), dtype=torch.float64, device=dev)
    cov = torch.empty((n_save, TRI, N), dtype=torch.float64, device=dev)
    ctx.bind_device(0, mean.data_ptr(), mean.numel() * 8)
    ctx.bind_device(1, cov.data_ptr(), cov.numel() * 8)
    ctx.set_problem_perturbed([1.0, 0.0, 0.0], [10.0, 28.0, 8.0 / 3.0], 0.0, 1e-2, first_index=rank * N)

    kernel_ms = []

    def one_pass():
        ctx.solve_fixed(tgrid)  # launches on torch's stream; records hipEvents around the kernel
        kernel_ms.append(ctx.kernel_time_ms(0)[0])
        if world > 1:
            return od.allgather_shards(mean[n_save - 1], world)
        return mean[n_save - 1]

    def sync_all():
        torch.cuda.synchronize(dev)
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        one_pass()
    kernel_ms.clear()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        final = one_pass()
    sync_all()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        el = float(t.item())

    # sanity inside the bench: the result is finite and the ensemble did something
    ok = bool(torch.isfinite(final).all().item()) and bool((ctx.get(10) == 0).all())

    total_steps = world * N * nsteps * args.steps
    value = total_steps / el
    k_ms = float(np.mean(kernel_ms))
    alg_bytes = B_ALG_STEP(D) * N * (nsteps + 1 if everystep else 1)
    achieved = alg_bytes / (k_ms * 1e-3)
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            key = f"lorenz63_ek1q3_N{N}_n{nsteps}_{args.save}"
            traffic = tj.get(key, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

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
            "scaling": "strong" if args.total_traj else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"Lorenz-63 d=3, EK1(order=3), dynamic diffusion, fixed dt=2^-9, {N} trajectories per GPU x {nsteps} steps, "
                            f"save={args.save} (BASELINE.json configs[2]; perturbed u0 via splitmix64)",
                "trajectories_per_gpu": N, "solver_steps": nsteps, "state_dim": D, "parallelism": f"ensemble-shard x{world}",
                "collective": "one all_gather of final means per pass" if world > 1 else "none",
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": achieved / HBM_PEAK, "traffic": traffic,
                "kernel": "odef::ek_filter_fixed_kernel<odef::RhsLorenz63, 3, true, %s, %s>" % (
                    "true" if everystep else "false", "true" if (everystep and N < 32768) else "false"),  # name as rocprofv3 prints it
                "kernel_ms": k_ms,
                "algorithmic_bytes_per_launch": alg_bytes,
            },
            "parity_ok": ok,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
            line["gpu_over_cpu"] = value / line["cpu_baseline"]["value"]
        print(json.dumps(line))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
