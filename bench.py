#!/usr/bin/env python3
"""Headline benchmark: simple_humanoid (free-flyer, nq 36 / nv 35) RNEA + ABA, fp64, batch 65536
per GPU (BASELINE.json configs[2]; the HIP counterpart of the reference's
benchmark/timings-parallel.cpp:151-207 which times rneaInParallel / abaInParallel).

  python bench.py --gpus N --steps K --warmup W
  python bench.py --gpus 8 --model talos_like --batch 1048576 --algo aba     # BASELINE.json configs[3]

One *step* = one rneaInParallel pass + one abaInParallel pass over the rank's batch (--algo picks
one of the two), inputs and outputs resident in HBM in the engine-native BATCH_MAJOR layout.  One
rank per GPU, no collective on the data path (instances are independent); ranks only meet in the
barriers around the timed region.  Rank 0 prints ONE JSON line.

Run without a launcher and with --gpus N > 1, this process starts the N ranks itself
(pinocchio_amd/launch.py: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* like torchrun) BEFORE anything
touches the GPU and stays GPU-free; under `python -m torch.distributed.run` it is one of the ranks.
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
    p.add_argument("--model", default="simple_humanoid",
                   help="built-in model name, or the path of a URDF file (free-flyer root is added)")
    p.add_argument("--algo", default="rnea+aba", choices=["rnea+aba", "rnea", "aba"])
    p.add_argument("--plugin", action="append", default=[],
                   help="kernel plugin(s) built by `python -m pinocchio_amd.codegen.plugin` for this model")
    p.add_argument("--repeats", type=int, default=9,
                   help="the --steps long timed region is run this many times; the line reports the median")
    p.add_argument("--armature", type=float, default=0.0,
                   help="rotor inertia on every 1-dof joint (model.armature; 0 = none, the reference benchmark's model)")
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
    n = min(q.shape[1], 65536)   # a bounded sample of the same workload
    qf, vf, af = (np.asfortranarray(x[:, :n]) for x in (q, v, a))
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
        "sample": f"{passes} passes of RNEA+ABA over the first {B} instances of the same batch, OpenMP static schedule",
        "rnea_evals_per_s": B * passes / t_r, "aba_evals_per_s": B * passes / t_a,
    }


def load_model(spec):
    from pinocchio_amd import sample_models, urdf
    if spec.endswith(".urdf") or os.sep in spec:
        with open(spec) as f:
            return urdf.build_model_from_urdf(f.read(), free_flyer=True, name=os.path.splitext(os.path.basename(spec))[0])
    return sample_models.build(spec)


def counted_flops(name):
    """Algorithmic flops per eval, counted with a counting scalar over the reference-structured
    restatement and frozen in BASELINE.md section 3 (tools/flops/count_flops.py)."""
    frozen = {"simple_humanoid": {"rnea": 10278, "aba": 27374}, "talos_like": {"rnea": 11295, "aba": 29936},
              "quadruped": {"rnea": 4515, "aba": 12856}, "arm7": {"rnea": 2277, "aba": 5633},
              "manipulator": {"rnea": 1938, "aba": 4779}, "baxter_simple": {"rnea": 6105, "aba": 15143}}
    return frozen.get(name[:-3] if name.endswith("_ff") else name)


def main():
    args = parse()
    from pinocchio_amd import launch
    if args.gpus > 1 and not launch.under_launcher():
        # one process per GPU, started before any GPU call; this parent never touches the device
        raise SystemExit(launch.spawn_ranks(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if os.environ.get("BRBD_BENCH_ECHO_RANK"):
        print(f"bench rank {rank}/{world}", file=sys.stderr, flush=True)
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

    for path in args.plugin:
        capi.load_compiled_kernels(path)
    model = load_model(args.model)
    if args.armature:
        for j in range(1, model.njoints):
            if model.nvs[j] == 1:
                model.armature[model.idx_vs[j]] = args.armature
    do_rnea, do_aba = "rnea" in args.algo, "aba" in args.algo
    B = args.batch
    pool = parallel.ModelPool(model, 1)
    qh, vh, ah = make_inputs(model, B, rank)
    dev = torch.device("cuda", local_rank)
    q, v, a = (torch.from_numpy(x).to(dev) for x in (qh, vh, ah))
    tau = torch.empty((model.nv, B), dtype=torch.float64, device=dev)
    ddq = torch.empty((model.nv, B), dtype=torch.float64, device=dev)
    dm = pool.device_model(local_rank)

    if not do_rnea:  # ABA alone: its torques come from one untimed RNEA pass
        parallel.rneaInParallel(1, pool, q, v, a, tau)
    evals_per_step = B * (int(do_rnea) + int(do_aba))

    def step():
        if do_rnea:
            parallel.rneaInParallel(1, pool, q, v, a, tau)
        if do_aba:
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
    # The timed region holds nothing but the calls (like the loop of benchmark/timings-parallel.cpp:101-112):
    # kernel timing is OFF in it.  `steps` steps between two fences, MAX over ranks; a region of 20
    # steps lasts 1 ms, so it is repeated `repeats` times and the line carries the median region.
    dm.set_timing(False)
    regions = []
    for _ in range(max(args.repeats, 1)):
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        regions.append(time.perf_counter() - t0)
    if dist is not None:
        t = torch.tensor(regions, dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        regions = [float(x) for x in t.tolist()]
    elapsed = float(np.median(regions))
    # Kernel durations for the roofline, in a region of their own AFTER the timed one: the same step
    # 1024 more times, ONE launch in 16 of each kernel carrying its own begin / end timestamps
    # (brbd_set_timing(m, 3), hipExtLaunchKernelGGL; the table-driven kernels between chained hipEvents
    # on the launch stream).  Timing events on every launch cost the stream 5-19 us per step and
    # lengthen the timed dispatches themselves (44.7 instead of 38 us for ABA), sampled ones 0.8 us
    # per step (tools/gap_check.py) -- either way they are the harness's time, not the path's, which
    # is why the timed region has none.
    k_timed, period = 1024, 16
    dm.kernel_time(capi.ALGO_RNEA)
    dm.kernel_time(capi.ALGO_ABA)
    dm.set_timing(True, dispatch=True, period=period)
    for _ in range(k_timed):
        step()
    fence()
    dm.set_timing(False)
    n_r, ms_r = dm.kernel_time(capi.ALGO_RNEA)
    n_a, ms_a = dm.kernel_time(capi.ALGO_ABA)
    if not do_rnea:
        n_r, ms_r = 0, 0.0

    # parity spot check of the timed outputs: ABA(RNEA(a)) == a (reference unittest/aba.cpp:143-154)
    err = float(((ddq - a).norm(dim=0) / a.norm(dim=0)).max().item()) if do_aba else None

    def timed_steps(fn, k):
        # (fresh buffers and kernels: the first few dozen launches run slower -- first-touch page
        # mappings, code-object load -- so warm up like the main region does; median of 3 regions)
        for _ in range(max(k // 2, 20)):
            fn()
        rates = []
        for _ in range(3):
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(k):
                fn()
            torch.cuda.synchronize()
            rates.append(evals_per_step * k / (time.perf_counter() - t))
        return float(np.median(rates))

    # Side figures, each in a region of its own, reported in `config`, never in `value`:
    # (each about 50 ms long whatever --steps is: a 5-step region measures the launch ramp, not the path)
    k_side = int(min(max(0.05 / (elapsed / args.steps), 50), 2000))
    # (1) the same step in the reference's own instance-major (Eigen column-major) layout, device-resident
    qi, vi, ai = (x.t().contiguous().t() for x in (q, v, a))
    taui = tau.t().contiguous().t()
    ddqi = torch.empty((B, model.nv), dtype=torch.float64, device=dev).t()

    def step_im():
        if do_rnea:
            parallel.rneaInParallel(1, pool, qi, vi, ai, taui)
        if do_aba:
            parallel.abaInParallel(1, pool, qi, vi, taui, ddqi)
    im_rate = timed_steps(step_im, k_side)
    # (2) the table-driven kernels (what a model unknown at build time runs on)
    variant = dm.compiled_kernels()
    td_rate = None
    if variant:
        dm.use_compiled(False)
        td_rate = timed_steps(step, max(k_side // 8, 3))
        dm.use_compiled(True)
    # (3) end to end from HOST memory in the reference's calling convention (Eigen column-major
    # matrices in, results back in host memory; PCIe-inclusive, SURVEY 8d "timing hygiene")
    host_rate = host_pinned_rate = None
    if rank == 0:
        def host_steps(qf, vf, af, tauf, outf):
            def step_host():
                if do_rnea:
                    parallel.rneaInParallel(1, pool, qf, vf, af, tauf)
                if do_aba:
                    parallel.abaInParallel(1, pool, qf, vf, tauf, outf)
            step_host()
            t1 = time.perf_counter()
            k_h = 0
            while k_h < 5 or (time.perf_counter() - t1 < 0.5 and k_h < 50):
                step_host()
                k_h += 1
            return evals_per_step * k_h / (time.perf_counter() - t1)

        tau_h = tau.cpu().numpy()
        # pageable memory (plain Eigen / numpy matrices): the copies block the host
        host_rate = host_steps(*(np.asfortranarray(x) for x in (qh, vh, ah, tau_h)), np.empty((model.nv, B), order="F"))
        # page-locked memory (hipHostMalloc / brbd_host_register): DMA copies, upload | kernels | download overlap
        def pinned(x):
            t = torch.empty((x.shape[1], x.shape[0]), dtype=torch.float64).pin_memory()
            t.copy_(torch.from_numpy(np.ascontiguousarray(x.T)))
            return t.numpy().T  # (dof, B) Fortran-ordered view of the pinned buffer
        host_pinned_rate = host_steps(pinned(qh), pinned(vh), pinned(ah), pinned(tau_h), pinned(np.zeros((model.nv, B))))

    if rank == 0:
        bytes_per_eval = (model.nq + 3 * model.nv) * 8   # q + v + a/tau in, tau/ddq out
        avg_r = ms_r / max(n_r, 1)
        avg_a = ms_a / max(n_a, 1)
        pm = {}
        tj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tj) and args.model == "simple_humanoid" and B == 65536 and not args.armature:
            try:
                with open(tj) as f:
                    pm = json.load(f)
            except Exception:
                pm = {}
        flops = counted_flops(model.name)
        executed = pm.get("executed_flops_per_eval") or {}
        # Dominant kernel: ABA when it runs, else RNEA.  Its binding roof is the lower of the two
        # ceilings in evals/s: HBM (8 TB/s over the algorithmic bytes per eval) and vector
        # FP64 (78.6 TFLOP/s over the flops the kernel EXECUTES per eval, PMC-counted).  The counted flops
        # of the reference-structured restatement (the WORLD-convention recursion)
        # are quoted beside it; the compiled ABA runs the reference's LOCAL-convention recursion
        # (aba.hxx:378-542), which needs 2.8x fewer flops for the same result, so a fraction of peak on
        # the WORLD count is no longer bounded by 1 and cannot serve as the roof.
        dom = "aba" if do_aba else "rnea"
        avg = avg_a if do_aba else avg_r
        kname = (f"brbd::spec_{variant}::k_{dom}<1, 1>" if variant else f"brbd::{dom}_kernel<1>")
        gbs = bytes_per_eval * B / (avg * 1e-3) / 1e9 if avg > 0 else 0.0
        traffic = pm.get(f"{dom}_kernel_bytes_per_launch")
        tsrc = ("static: profiles/pmc_traffic.json (rocprofv3 --pmc passes of this workload, tools/pmc_passes.sh + "
                "tools/pmc_traffic.py), not measured in this run") if traffic else None
        ex = executed.get(dom)
        tf_ex = ex * B / (avg * 1e-3) / 1e12 if (ex and avg > 0) else None
        tf_counted = flops[dom] * B / (avg * 1e-3) / 1e12 if (flops and avg > 0) else None
        hbm_ceiling = HBM_PEAK_GBS * 1e9 / bytes_per_eval
        fp64_ceiling = FP64_PEAK_TFLOPS * 1e12 / ex if ex else None
        fp64 = {"peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                "executed_flops_per_eval": ex, "achieved_executed": tf_ex,
                "frac_executed": tf_ex / FP64_PEAK_TFLOPS if tf_ex else None,
                "counted_flops_per_eval_reference_world_recursion": flops[dom] if flops else None,
                "achieved_counted": tf_counted, "frac_counted": tf_counted / FP64_PEAK_TFLOPS if tf_counted else None,
                "flops_source": "executed: SQ_INSTS_VALU_{FMA,ADD,MUL}_F64 per wave (profiles/pmc_traffic.json); counted: a counting "
                                "scalar over the reference-structured restatement (BASELINE.md section 3, profiles/counted_flops.json)",
                "ceiling_evals_per_s": fp64_ceiling}
        hbm = {"achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
               "algorithmic_bytes_per_launch": bytes_per_eval * B, "ceiling_evals_per_s": hbm_ceiling}
        if fp64_ceiling is not None and fp64_ceiling < hbm_ceiling:
            roof = {"bound": "fp64", "kernel": kname, "achieved": tf_ex, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": tf_ex / FP64_PEAK_TFLOPS, "algorithmic_flops_per_launch": ex * B, "traffic": traffic,
                    "traffic_source": tsrc, "fp64": fp64, "hbm": hbm}
        else:
            roof = {"bound": "hbm", "kernel": kname, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": gbs / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": bytes_per_eval * B, "traffic": traffic,
                    "traffic_source": tsrc, "fp64": fp64, "hbm": hbm}
        if fp64["frac_counted"] is not None:
            roof["note"] = ("frac prices the kernel AS BUILT against its binding roof (the lower ceiling in evals/s of HBM over the "
                            "algorithmic bytes and FP64 over the flops it executes). On SURVEY 8d's algorithmic flops (counted on the "
                            "reference-structured WORLD recursion, BASELINE.md section 3) the same launch is at fp64.frac_counted = "
                            "%.2f of the FP64 peak: the compiled kernel runs the reference's LOCAL recursion and does %.1fx less "
                            "arithmetic for the same result." % (fp64["frac_counted"], (flops[dom] / ex) if ex else float("nan")))
        headline = args.model == "simple_humanoid" and B == 65536 and args.algo == "rnea+aba" and not args.armature
        out = {
            "metric": ("ABA/RNEA evals/sec, humanoid fp64 batch 65536" if headline
                       else f"{args.algo} evals/sec, {model.name} fp64 batch {B} per GPU"),
            "value": evals_per_step * args.steps * world / elapsed,
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
                "workload": f"{model.name} (nq {model.nq} / nv {model.nv}) {args.algo.upper().replace('+', ' + ')}, "
                            f"batch {B} per GPU, fp64, device-resident BATCH_MAJOR",
                "batch_per_gpu": B,
                "armature": args.armature,
                "evals_per_step_per_gpu": evals_per_step,
                "rnea_kernel_ms": avg_r if do_rnea else None, "aba_kernel_ms": avg_a if do_aba else None,
                "repeats": len(regions),
                "region_ms": {"median": elapsed * 1e3, "min": min(regions) * 1e3, "max": max(regions) * 1e3},
                "kernel_timing": {"how": "separate region after the timed one (no event inside the timed region): "
                                         f"{k_timed} more steps, one launch in {period} of each kernel with the dispatch's own "
                                         "begin / end timestamps (hipExtLaunchKernelGGL events on the launch stream)",
                                  "period": period,
                                  "timed_launches": {"rnea": n_r, "aba": n_a}},
                "rnea_evals_per_s_kernel": B / (avg_r * 1e-3) if avg_r > 0 else None,
                "aba_evals_per_s_kernel": B / (avg_a * 1e-3) if avg_a > 0 else None,
                "instance_major_evals_per_s": im_rate,
                "table_driven_evals_per_s": td_rate,
                "host_instance_major_evals_per_s": host_rate,
                "host_pinned_instance_major_evals_per_s": host_pinned_rate,
                "max_rel_err_aba_of_rnea": err,
                "kernels": (f"model-compiled ({variant})" if variant else "table-driven"),
                "parallelism": f"dp{world} (independent slices, no collective)",
            },
            "roofline": roof,
        }
        if do_rnea and do_aba:
            gr = bytes_per_eval * B / (avg_r * 1e-3) / 1e9 if avg_r > 0 else 0.0
            out["roofline_rnea"] = {
                "bound": "hbm", "kernel": kname.replace("k_aba", "k_rnea").replace("aba_kernel", "rnea_kernel"),
                "achieved": gr, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gr / HBM_PEAK_GBS,
                "traffic": pm.get("rnea_kernel_bytes_per_launch"),
                "fp64_frac_counted": (flops["rnea"] * B / (avg_r * 1e-3) / 1e12 / FP64_PEAK_TFLOPS) if flops and avg_r > 0 else None,
                "fp64_frac_executed": (executed["rnea"] * B / (avg_r * 1e-3) / 1e12 / FP64_PEAK_TFLOPS) if executed.get("rnea") and avg_r > 0 else None,
            }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model, qh, vh, ah, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
