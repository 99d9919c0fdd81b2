"""World-size-2 test of the ensemble sharding + the single all-gather (gloo on CPU).
The per-shard "result" here is the oracle's splitmix64 ensemble (tests may use the oracle);
on the GPU box the same code path carries the kernels' final means over RCCL."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "oracle"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import odefilters_jl_amd  # noqa: F401
    from odefilters_jl_amd import dist as od
    import odefilter_oracle as orc

    r, w, _ = od.init_from_env(backend="gloo")
    lo, hi = od.shard_bounds(total, r, w)
    local = torch.from_numpy(orc.ensemble_u0(np.array([1.0, 0.0, 0.0]), hi - lo, 1e-2, first=lo).T.copy())  # [d, n_local]
    g = od.allgather_shards(local, w)
    full = od.gathered_to_global(g)
    if r == 0:
        q.put(full.numpy())
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_shard_bounds(pkg):
    from odefilters_jl_amd import dist as od

    for total, world in ((65536, 8), (10, 3), (7, 8)):
        b = [od.shard_bounds(total, r, world) for r in range(world)]
        assert b[0][0] == 0 and b[-1][1] == total
        assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
        assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


def test_two_rank_allgather_reassembles_ensemble(orc):
    total, world = 64, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    full = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    np.testing.assert_array_equal(full.T, orc.ensemble_u0(np.array([1.0, 0.0, 0.0]), total, 1e-2))


def _worker_solution_gather(rank, world, port, total, q):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "oracle"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import odefilters_jl_amd as pkg
    from odefilters_jl_amd import dist as od
    import odefilter_oracle as orc

    od.init_from_env(backend="gloo")
    ens = pkg.EnsembleProblem(pkg.ODEProblem("lorenz63", [1.0, 0.0, 0.0], (0.0, 1.0), [10.0, 28.0, 8.0 / 3.0]), perturb_scale=1e-2)
    local, n_local, (lo, hi) = pkg.shard_ensemble(ens, total, rank, world)
    assert local.first_index == lo and n_local == hi - lo

    class ShardResult(pkg.EnsembleSolution):  # the solve itself needs a GPU; its final means are stood in for by u0
        def __init__(self):
            self.D, self.shard = 3, (lo, hi, total, world)

        def final_mean(self):
            return orc.ensemble_u0(np.array(local.prob.u0), n_local, local.perturb_scale, seed=local.seed, first=local.first_index)

    full = ShardResult().gather_final()
    q.put((rank, full))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_distributed_solution_gather_with_uneven_shards(orc):
    """EnsembleHIP(distributed=True): shard_ensemble keeps the global numbering of the synthetic ensemble and
    EnsembleSolution.gather_final reassembles [N_total, D] in global order on every rank (7 trajectories on 2 ranks)."""
    total, world = 7, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_solution_gather, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want = orc.ensemble_u0(np.array([1.0, 0.0, 0.0]), total, 1e-2)
    for r in range(world):
        np.testing.assert_array_equal(got[r], want)


def test_shard_ensemble_slices_explicit_inputs(pkg):
    u0s = np.arange(30.0).reshape(10, 3)
    ps = np.arange(10.0).reshape(10, 1)
    ens = pkg.EnsembleProblem(pkg.ODEProblem("lorenz63", [1.0, 0.0, 0.0], (0.0, 1.0), [10.0, 28.0, 8.0 / 3.0]), u0s=u0s, ps=ps)
    parts = [pkg.shard_ensemble(ens, None, r, 4) for r in range(4)]
    np.testing.assert_array_equal(np.concatenate([p[0].u0s for p in parts]), u0s)
    np.testing.assert_array_equal(np.concatenate([p[0].ps for p in parts]), ps)
    assert [p[1] for p in parts] == [3, 3, 2, 2]
