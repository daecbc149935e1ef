"""-m gpu: the sharded MPPI path with the REAL engine and a real process group — two processes on the
one GPU of the test box, gloo backend on CUDA tensors (RCCL refuses two ranks on one device; the
collective call site is the same `all_gather_into_tensor`).  Result must equal one handle of 2x the size."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, N_local, H, p, steps, seed, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from control_toolkit_amd import CtkEngine
    from control_toolkit_amd.dist import ShardedMPPI
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = CtkEngine("mppi", "ODE", num_rollouts=N_local, mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=p,
                    seed=seed, global_rollout_offset=rank * N_local)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    sh = ShardedMPPI(eng, rank, world, device=torch.device("cuda", 0))
    s = np.array([0.05, -0.1, 2.8, 0.4], np.float32)
    us = []
    for t in range(steps):
        torch.cuda.synchronize()
        us.append(float(sh.step(s, None)[0]))        # device Philox addressed by GLOBAL rollout index
    q.put((rank, us, eng.read("U_NOM")))
    dist.barrier()
    eng.close()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_process_sharded_mppi_equals_single_handle():
    from control_toolkit_amd import CtkEngine
    N, H, p, steps, seed = 2048, 40, 10, 3, 77
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, N // 2, H, p, steps, seed, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda r: r[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    full = CtkEngine("mppi", "ODE", num_rollouts=N, mpc_horizon=H, dt=0.02, period_interpolation_inducing_points=p, seed=seed)
    s = np.array([0.05, -0.1, 2.8, 0.4], np.float32)
    ref = [float(full.step(s, None)[0]) for _ in range(steps)]
    np.testing.assert_array_equal(res[0][1], res[1][1])                       # identical on both ranks
    np.testing.assert_allclose(res[0][1], ref, rtol=1e-4, atol=2e-5)          # == unsharded (same global draws)
    np.testing.assert_allclose(res[0][2], full.read("U_NOM"), rtol=1e-4, atol=2e-5)
    full.close()
