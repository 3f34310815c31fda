"""Multi-GPU path rehearsed on CPU: world_size 2, gloo.  The path shards trivially (independent
instances): each rank owns the contiguous column range [r*B/G, (r+1)*B/G) and there is
no collective on the data path; only the timing barrier and the MAX-reduce of bench.py are
exercised here, with the oracle standing in for the device (the product path has no CPU mode).
The ranks are started by pinocchio_amd/launch.py: spawn_ranks, the launcher `bench.py --gpus N`
itself uses when it is not already running under torchrun."""
import os
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


def _run_through_launcher(tmp_path, script_text, world):
    """Runs `world` ranks of a worker through launch.spawn_ranks in a GPU-free parent process, the
    way `python bench.py --gpus N` does, and returns (exit code, stdout of rank 0, stderr)."""
    script = tmp_path / "worker.py"
    script.write_text(script_text)
    parent = tmp_path / "parent.py"
    parent.write_text(textwrap.dedent("""
        import sys
        sys.path.insert(0, %r)
        from pinocchio_amd import launch
        assert "torch" not in sys.modules  # the parent stays torch- and GPU-free
        raise SystemExit(launch.spawn_ranks(%d, [%r], timeout=280))
    """) % (ROOT, world, str(script)))
    env = dict(os.environ, OMP_NUM_THREADS="2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(parent)], env=env, capture_output=True, text=True, timeout=300)
    return r.returncode, r.stdout, r.stderr


def test_two_rank_sharding(tmp_path):
    rc, out, err = _run_through_launcher(tmp_path, WORKER % (ROOT, ROOT), 2)
    assert rc == 0, out + err
    assert "OK" in out


def test_launcher_reports_a_failed_rank(tmp_path):
    rc, out, err = _run_through_launcher(
        tmp_path, "import os, sys\nsys.exit(3 if os.environ['RANK'] == '1' else 0)\n", 2)
    assert rc == 3


def test_bench_gpus_flag_spawns_ranks_before_touching_the_gpu(tmp_path):
    """`python bench.py --gpus 2` outside torchrun must start 2 ranks (here they stop at the
    'needs a GPU' check, which proves WORLD_SIZE reached them) and propagate the failure."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["BRBD_BENCH_ECHO_RANK"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert "bench rank 0/2" in r.stderr and "bench rank 1/2" in r.stderr, r.stderr
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0 and "needs a GPU" in r.stderr


def test_column_slices_partition_the_batch():
    from pinocchio_amd import launch
    for B, G in ((65536, 8), (1000, 3), (7, 8)):
        edges = [launch.column_slice(B, r, G) for r in range(G)]
        assert edges[0][0] == 0 and edges[-1][1] == B
        assert all(edges[i][1] == edges[i + 1][0] for i in range(G - 1))
