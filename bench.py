#!/usr/bin/env python3
"""Headline benchmark: simple_humanoid (free-flyer, nq 36 / nv 35) RNEA + ABA, fp64, batch 65536
per GPU (BASELINE.json configs[2]; the HIP counterpart of the reference's
benchmark/timings-parallel.cpp:151-207 which times rneaInParallel / abaInParallel).

  python bench.py --gpus N --steps K --warmup W

One *step* = one rneaInParallel pass + one abaInParallel pass over the rank's batch, inputs and
outputs resident in HBM in the engine-native BATCH_MAJOR layout.  One rank per GPU, no collective
on the data path (instances are independent); ranks only meet in the barriers around the timed
region.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_PEAK_TFLOPS = 78.6    # MI355X vector FP64 (spec; no MFMA on this path)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=2000)
    p.add_argument("--warmup", type=int, default=300)
    p.add_argument("--batch", type=int, default=65536, help="instances per GPU")
    p.add_argument("--model", default="simple_humanoid")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=12.0)
    return p.parse_args()


def make_inputs(model, B, rank):
    """Seeded synthetic inputs mirroring benchmark/timings-parallel.cpp:48-61: q random
    configuration (normalised quaternion), v, a ~ U[-1,1].  Rank r draws slice r of the stream."""
    rng = np.random.default_rng([0, rank])
    q = np.ascontiguousarray(model.random_configuration(rng, B))  # (nq, B) C-order = BATCH_MAJOR
    v = rng.uniform(-1, 1, (model.nv, B))
    a = rng.uniform(-1, 1, (model.nv, B))
    return q, v, a


def cpu_baseline(model, q, v, a, seconds):
    """The oracle's OpenMP batch loops (CPU restatement of rneaInParallel / abaInParallel;
    the reference itself cannot be built offline) on the host cores of this box."""
    import oracle
    O = oracle.Oracle(model)
    threads = oracle.max_threads()
    qf, vf, af = (np.asfortranarray(x) for x in (q, v, a))
    B = qf.shape[1]
    tau = O.rnea(qf, vf, af, threads=threads)   # warm-up (thread pool, page faults)
    O.aba(qf, vf, tau, threads=threads)
    t_r = t_a = 0.0
    passes = 0
    t0 = time.perf_counter()
    while True:
        t = time.perf_counter()
        tau = O.rnea(qf, vf, af, threads=threads)
        t_r += time.perf_counter() - t
        t = time.perf_counter()
        O.aba(qf, vf, tau, threads=threads)
        t_a += time.perf_counter() - t
        passes += 1
        if time.perf_counter() - t0 >= seconds or passes >= 200:
            break
    return {
        "value": 2 * B * passes / (t_r + t_a), "unit": "evals/s", "cores": threads, "kind": "port",
        "sample": f"{passes} passes of RNEA+ABA over the same {B}-instance batch, OpenMP static schedule",
        "rnea_evals_per_s": B * passes / t_r, "aba_evals_per_s": B * passes / t_a,
    }


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    from pinocchio_amd import capi, parallel, sample_models

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (torch.cuda.is_available() is False)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    model = sample_models.build(args.model)
    B = args.batch
    pool = parallel.ModelPool(model, 1)
    qh, vh, ah = make_inputs(model, B, rank)
    dev = torch.device("cuda", local_rank)
    q, v, a = (torch.from_numpy(x).to(dev) for x in (qh, vh, ah))
    tau = torch.empty((model.nv, B), dtype=torch.float64, device=dev)
    ddq = torch.empty((model.nv, B), dtype=torch.float64, device=dev)
    dm = pool.device_model(local_rank)

    def step():
        parallel.rneaInParallel(1, pool, q, v, a, tau)
        parallel.abaInParallel(1, pool, q, v, tau, ddq)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # the GPU's clocks take a few hundred ms of load to settle (step time 75 -> 68 us): run the step
    # for 0.3 s before the W warm-up steps, so that short --steps / --warmup runs measure the same
    # machine state as long ones
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 0.3:
        for _ in range(50):
            step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    fence()
    dm.kernel_time(capi.ALGO_RNEA)
    dm.kernel_time(capi.ALGO_ABA)
    # hipEvents between the kernel launches, on the launch stream (chained: the event that closes
    # one launch opens the next, one event per launch)
    dm.set_timing(True, chained=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    dm.set_timing(False)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    n_r, ms_r = dm.kernel_time(capi.ALGO_RNEA)
    n_a, ms_a = dm.kernel_time(capi.ALGO_ABA)

    # parity spot check of the timed outputs: ABA(RNEA(a)) == a (reference unittest/aba.cpp:143-154)
    err = float(((ddq - a).norm(dim=0) / a.norm(dim=0)).max().item())

    # the same step with the reference's own instance-major (Eigen column-major) layout, untimed
    # region of its own: reported in config, never in `value`
    qi, vi, ai = (x.t().contiguous().t() for x in (q, v, a))
    taui = torch.empty((B, model.nv), dtype=torch.float64, device=dev).t()
    ddqi = torch.empty((B, model.nv), dtype=torch.float64, device=dev).t()
    for _ in range(3):
        parallel.rneaInParallel(1, pool, qi, vi, ai, taui)
        parallel.abaInParallel(1, pool, qi, vi, taui, ddqi)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    k_im = max(args.steps // 4, 5)
    for _ in range(k_im):
        parallel.rneaInParallel(1, pool, qi, vi, ai, taui)
        parallel.abaInParallel(1, pool, qi, vi, taui, ddqi)
    torch.cuda.synchronize()
    im_rate = 2 * B * k_im / (time.perf_counter() - t1)

    if rank == 0:
        bytes_per_eval = (model.nq + 3 * model.nv) * 8   # q + v + a/tau in, tau/ddq out
        avg_r = ms_r / max(n_r, 1)
        avg_a = ms_a / max(n_a, 1)
        # dominant kernel = ABA
        ach = bytes_per_eval * B / (avg_a * 1e-3) / 1e9 if avg_a > 0 else 0.0
        traffic = None
        tj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tj):
            try:
                with open(tj) as f:
                    traffic = json.load(f).get("aba_kernel_bytes_per_launch")
            except Exception:
                traffic = None
        # algorithmic flops per eval (SURVEY 8d pre-estimates); what the kernels execute is in
        # profiles/pmc_traffic.json (PMC: SQ_INSTS_VALU_{FMA,ADD,MUL}_F64, FMA = 2)
        flops = {"rnea": 8.5e3, "aba": 18.5e3}
        executed = None
        try:
            with open(tj) as f:
                executed = json.load(f).get("executed_flops_per_eval")
        except Exception:
            executed = None
        variant = dm.compiled_kernels()
        kname = (f"brbd::spec_{variant}::k_aba<1, 1>" if variant else "brbd::aba_kernel<1>")
        out = {
            "metric": "ABA/RNEA evals/sec, humanoid fp64 batch 65536",
            "value": 2 * B * args.steps * world / elapsed,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{args.model} (free-flyer, nq {model.nq} / nv {model.nv}) RNEA + ABA, "
                            f"batch {B} per GPU, fp64, device-resident BATCH_MAJOR",
                "batch_per_gpu": B,
                "evals_per_step_per_gpu": 2 * B,
                "rnea_kernel_ms": avg_r, "aba_kernel_ms": avg_a,
                "rnea_evals_per_s_kernel": B / (avg_r * 1e-3) if avg_r > 0 else None,
                "aba_evals_per_s_kernel": B / (avg_a * 1e-3) if avg_a > 0 else None,
                "instance_major_evals_per_s": im_rate,
                "max_rel_err_aba_of_rnea": err,
                "kernels": (f"model-compiled ({variant})" if variant else "table-driven"),
                "parallelism": f"dp{world} (independent slices, no collective)",
            },
            "roofline": {
                "bound": "hbm", "kernel": kname, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                "algorithmic_bytes_per_launch": bytes_per_eval * B,
            },
            "roofline_fp64": {
                "kernel": kname, "achieved": flops["aba"] * B / (avg_a * 1e-3) / 1e12 if avg_a > 0 else None,
                "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": (flops["aba"] * B / (avg_a * 1e-3) / 1e12 / FP64_PEAK_TFLOPS) if avg_a > 0 else None,
                "rnea_achieved": flops["rnea"] * B / (avg_r * 1e-3) / 1e12 if avg_r > 0 else None,
                "algorithmic_flops_per_eval": flops,
                "executed_flops_per_eval": executed,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model, qh, vh, ah, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
