"""Multi-GPU path rehearsed on CPU: world_size 2, gloo.  The path shards trivially (independent
instances): each rank owns the contiguous column range [r*B/G, (r+1)*B/G) and there is
no collective on the data path; only the timing barrier and the MAX-reduce of bench.py are
exercised here, with the oracle standing in for the device (the product path has no CPU mode)."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, time
    import numpy as np
    import torch, torch.distributed as dist
    sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tests"))
    import oracle
    from pinocchio_amd import sample_models as S
    import bench
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    m = S.quadruped()
    B = 64
    # same generator bench.py uses: rank r draws slice r
    q, v, a = bench.make_inputs(m, B, rank)
    O = oracle.Oracle(m)
    dist.barrier()
    t0 = time.perf_counter()
    tau = O.rnea(q, v, a)
    ddq = O.aba(q, v, tau)
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.barrier()
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    # gather only to CHECK the slices (never on the timed path)
    outs = [torch.zeros((m.nv, B), dtype=torch.float64) for _ in range(world)]
    dist.all_gather(outs, torch.from_numpy(np.ascontiguousarray(ddq)))
    if rank == 0:
        full = np.concatenate([o.numpy() for o in outs], axis=1)
        ref = []
        for r in range(world):
            qr, vr, ar = bench.make_inputs(m, B, r)
            ref.append(O.aba(qr, vr, O.rnea(qr, vr, ar)))
        ref = np.concatenate(ref, axis=1)
        assert np.array_equal(full, ref)
        # slices differ between ranks (distinct streams) and the union is rank-count independent
        assert not np.array_equal(outs[0].numpy(), outs[1].numpy())
        print("OK", float(el))
    dist.destroy_process_group()
""")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_sharding(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % (ROOT, ROOT))
    env = dict(os.environ, OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "OK" in r.stdout
