#!/usr/bin/env python3
"""bench.py -- headline benchmark of the GP hot path on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic input (config C2):
    ARD-RBF Gram (n=8192, d=8) -> fp64 blocked Cholesky -> alpha, LML  (GpPredictor.preComputeComponents)
    -> posterior mean + variance at m=65536 test points                  (GpPredictor.predict, diag variance)
with X, y, X* already resident in HBM.  value = test points / second over the whole job.

N > 1, one rank per GPU.  Two ways in, same ranks either way:
  * under an external launcher (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`): RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* come from the environment;
  * plain `python bench.py --gpus N` with no WORLD_SIZE in the environment: this process starts the N ranks itself as child
    processes (self_launch) BEFORE anything touches the GPU, never initialises the GPU itself and exits with their status.
The path shards over test points -- every rank fits the same model redundantly (no data-path collective; SURVEY.md 8e: refit
7 ms vs broadcasting 0.5 GB of L) and predicts its own m points; weak scaling, value = N*m / max-rank time; the posterior of all
N*m points is assembled on every rank by one all_gather per step.  The same JSON line also carries the north-star's own scaling
workload as `c3_sharded`: LML + gradient over 64 settings at n = 4096 sharded 64/N per GPU (strong scaling, settings/s, per-rank
seconds, and the C-ABI's own RCCL path gp_dist_lml_grad_batched checked against it).
torch is used for rendezvous/barrier/max-reduce only; all compute goes through libgpcore.so.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP64_MFMA_TFLOPS = 78.6   # MI355X vendor fp64 matrix peak; the probe value is printed next to it
PEAK_HBM_GBS = 8000.0


def progress(msg):
    """One line on stderr when GPCORE_BENCH_PROGRESS is set: under a profiler that may stop the program (a PMC pass cut by its time
    limit leaves no trace of how far the program got) the log then says where."""
    if os.environ.get("GPCORE_BENCH_PROGRESS"):
        print("[bench %.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def host_cores():
    """Cores this process may really use: the affinity mask, cut by a cgroup CPU quota when there is one (the GPU box hands a
    16-core share of a 256-thread host: spawning one BLAS thread per visible CPU only oversubscribes it)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("GPCORE_CPU_THREADS", "64"))))


def cpu_baseline(p, L_host, alpha_host, m_total, predict_pts=32, fit_full_size=False):
    """The stated CPU baseline (`cpu_ref`): oracle/gp_oracle.c -- the 1-thread restatement of the reference's
    loops -- timed on a BOUNDED sample of the same workload.  Fit (Gram + unblocked Cholesky + two substitution solves,
    GpPredictor.preComputeComponents) is timed at n = 1024, 2048, 4096 on leading subsets of the same X, y and extrapolated to
    the full n by the power law fitted through ALL THREE sizes (least squares in log-log; the exponent of the last doubling alone
    is printed beside it: the row-walking substitution and the unblocked Cholesky fall out of cache as n grows, so the growth per
    doubling rises -- 7x then 13x measured -- and the two exponents bracket n = 8192).  `--cpu-fit-full` times the fit at the
    full n instead (about a minute).  Predict (cross-Gram + scalar forward substitution + variance per point) is timed at the
    full n against the real factor on `predict_pts` = 32 of the test points (about 1.7 s each: the substitution walks the rows of a
    column-major 8192 x 8192 factor).  value = points/s of a whole fit + predict step, the benchmark's metric; the predict term
    is 99.9 % of it."""
    from oracle import gp_oracle as orc
    orc.build()
    n = p["X"].shape[0]
    sizes = [s_ for s_ in (1024, 2048, 4096) if s_ <= n]
    if fit_full_size and n not in sizes:
        sizes.append(n)
    fit_s = []
    for s_ in sizes:
        Xs_, ys_ = np.asfortranarray(p["X"][:s_]), np.ascontiguousarray(p["y"][:s_])
        t0 = time.perf_counter()
        orc.fit(Xs_, ys_, p["theta"])
        fit_s.append(time.perf_counter() - t0)
    expo_last = float(np.log(fit_s[-1] / fit_s[-2]) / np.log(sizes[-1] / sizes[-2])) if len(sizes) >= 2 else 3.0
    if len(sizes) >= 3:
        expo, icpt = (float(v) for v in np.polyfit(np.log(sizes[-3:]), np.log(fit_s[-3:]), 1))
    else:
        expo, icpt = expo_last, float(np.log(fit_s[-1]) - expo_last * np.log(sizes[-1]))
    if sizes and sizes[-1] == n:
        fit_full, extrapolated = fit_s[-1], False
    else:
        fit_full, extrapolated = float(np.exp(icpt + expo * np.log(float(n)))), True
    k = int(max(2, min(predict_pts, p["Xs"].shape[0])))
    xs = np.asfortranarray(p["Xs"][:k])
    t0 = time.perf_counter()
    mean, var, _, _ = orc.predict(p["X"], p["theta"], L_host, alpha_host, xs)
    dt = time.perf_counter() - t0
    step_s = fit_full + m_total * dt / k
    return dict(value=m_total / step_s, unit="points/s", cores=1, kind="port", extrapolated=extrapolated,
                fit_s_measured={str(s_): t for s_, t in zip(sizes, fit_s)}, fit_growth_exponent_three_sizes=expo,
                fit_growth_exponent_last_doubling=expo_last,
                fit_s_at_n=fit_full, predict_points_per_s=k / dt, predict_points_timed=k, step_s_at_n=step_s,
                sample="oracle/gp_oracle.c, 1 thread: fit timed at n=%s on leading subsets of the same X, y%s; predict timed on %d of "
                       "the %d test points at the full n=%d against the factor of the GPU fit; value = %d points / (fit + %d "
                       "points at the measured rate)"
                       % ("/".join(map(str, sizes)),
                          " and extrapolated to n=%d by the power law fitted through the three sizes (exponent %.2f; last doubling "
                          "alone %.2f)" % (n, expo, expo_last)
                          if extrapolated else " (the full n included: nothing extrapolated)",
                          k, m_total, n, m_total, m_total)), mean, var, k


def _cholesky_block(n, t_fit, t_fit_serial, o_k, o_ms, o_work, p_k, p_ms, p_work):
    """Cholesky figures of the bench line.  The trailing update is reported over ALL update launches -- the K = 512 outer updates
    and the K = 128 in-panel updates -- with SURVEY.md 8(d)'s exact count  sum_k nb r_k (r_k + 1),  r_k = n - (k + 1) nb,  nb = 128
    (the kernels' own tile count, which includes the strict upper half of the diagonal tiles, is given beside it)."""
    nb = 128
    exact = float(sum(nb * (n - (k + 1) * nb) * (n - (k + 1) * nb + 1) for k in range(n // nb - 1)))
    upd_s = (o_ms + p_ms) * 1e-3
    out = {"fit_ms": t_fit * 1e3, "total_tflops": (n ** 3 / 3.0) / t_fit / 1e12,
           "fit_ms_lookahead_off": t_fit_serial * 1e3, "panel_width_inner": nb, "outer_update_K": 512,
           "trailing_update_tflops": exact / upd_s / 1e12 if upd_s > 0 else 0.0,
           "trailing_update_frac_of_peak": exact / upd_s / 1e12 / PEAK_FP64_MFMA_TFLOPS if upd_s > 0 else 0.0,
           "trailing_update_flops_exact": exact, "trailing_update_flops_tiles": o_work + p_work,
           "trailing_update_ms": (o_ms + p_ms), "trailing_update_launches": o_k + p_k,
           "outer_update_tflops_tiles": o_work / (o_ms * 1e-3) / 1e12 if o_ms > 0 else 0.0, "outer_update_launches": o_k,
           "in_panel_update_tflops_tiles": p_work / (p_ms * 1e-3) / 1e12 if p_ms > 0 else 0.0, "in_panel_update_launches": p_k,
           "trailing_update_measured": "one refit with the look-ahead switched off (gp_ctx_set_lookahead 0), HIP events per launch"}
    return out


PMC_TAG = "r04"          # profiles/<PMC_TAG>_pmc_<workload>_summary.json: the committed rocprofv3 --pmc passes of this round (tools/collect_evidence.sh)
CLASS_SYMBOLS = {        # kernel symbols behind a profile class, as tools/pmc_summary.py abbreviates them
    "gemm": ("gemm_nt_f64_kernel<0,", "gemm_fused_kernel<"),
    "syrk": ("gemm_nt_f64_kernel<1,", "gemm_k128_kernel<1>"),
    "panel": ("gemm_nt_f64_kernel<1,1", "gemm_k128_kernel<1>"),
}


def kernel_sources_sha256():
    """Hash of the library's sources as tools/pmc_summary.py records it in a PMC summary."""
    import glob
    import hashlib
    root = os.path.join(ROOT, "gp_algos_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(root, "*.hip")) + glob.glob(os.path.join(root, "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def _pmc_traffic(workload, prefixes):
    """HBM bytes per launch of the kernels behind a profile class, from the committed PMC passes of the same bench command
    (FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE, MI355X_MICROARCH.md): launch-weighted mean over the matching
    symbols.  PMC counters cannot be read from inside the process: None when the summary file is missing, or when it was taken on
    other kernel sources than the ones in this tree (the summary carries their hash)."""
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "%s_pmc_%s_summary.json" % (PMC_TAG, workload))))
    except Exception:
        return None
    # counters of another build are not this code's counters: refused, `traffic` is then null
    if pm.get("kernel_sources_sha256") != kernel_sources_sha256():
        return None
    tot, cnt, used = 0.0, 0, []
    for sym, e in pm.get("kernels", {}).items():
        if not sym.startswith(tuple(prefixes)) or "hbm_bytes_per_launch_corrected" not in e:
            continue
        k = int(e.get("launches_FETCH_SIZE", 0))
        tot += e["hbm_bytes_per_launch_corrected"] * k
        cnt += k
        used.append(sym)
    if not cnt:
        return None
    return {"bytes_per_launch": tot / cnt, "launches_in_pmc_run": cnt, "symbols": sorted(used),
            "source": "profiles/%s_pmc_%s_summary.json" % (PMC_TAG, workload)}


def _roofline_from_profile(ctx, L, classes, names, workload=None):
    """The kernel class that took the most stream time during the timed steps (HIP events recorded by the library around every
    launch of that class, on the stream the launch went to): achieved = algorithmic flops of its launches / their summed time."""
    best = None
    for cls in classes:
        k, ms, work = ctx.profile_read(cls)
        if k and ms > 0 and (best is None or ms > best[2]):
            best = (cls, k, ms, work)
    if best is None:
        return None
    cls, k, ms, work = best
    tf = work / (ms * 1e-3) / 1e12
    key = {L.GP_PROF_GEMM: "gemm", L.GP_PROF_SYRK: "syrk", L.GP_PROF_PANEL_UPD: "panel"}.get(cls)
    pmc = _pmc_traffic(workload, CLASS_SYMBOLS[key]) if (workload and key) else None
    out = {"kernel": names[cls], "bound": "mfma", "achieved": tf, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
           "frac": tf / PEAK_FP64_MFMA_TFLOPS, "traffic": pmc["bytes_per_launch"] if pmc else None, "launches": k,
           "avg_launch_us": ms / k * 1e3, "flops_per_launch_avg": work / k, "class_time_ms": ms}
    if pmc:
        # an MFMA-bound product C -= A B^T on 128 x 128 tiles moves at least 16 B of C per 2 K flops per element: what the counters
        # add to that is operand re-reads
        out["traffic_unit"] = "HBM bytes per launch (rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE), launch-weighted over %s; %s" % (pmc["symbols"], pmc["source"])
        out["flops_per_hbm_byte"] = (work / k) / pmc["bytes_per_launch"]
    return out


def run_secondary(args):
    """Secondary workloads (not the headline metric): C3 batched LML+gradient sharded over ranks with one all_gather of the
    results, C4 EP sweeps (replicas only), C5 one 10^6-point posterior request split over the ranks."""
    import torch
    from gp_algos_amd import dist as gdist
    rank, local_rank, world = gdist.env_rank_world()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
    # Rehearsal knobs for a 1-GPU box only (never set by the driver): several ranks on one device over gloo.
    if os.environ.get("GPCORE_BENCH_DEVICE") is not None:
        local_rank = int(os.environ["GPCORE_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    backend = os.environ.get("GPCORE_BENCH_BACKEND", "nccl")
    gdist.init(backend, torch.device("cuda", local_rank))
    coll_dev = "cuda" if backend == "nccl" else "cpu"
    import __graft_entry__ as entry
    entry.build()
    from gp_algos_amd import _lib as L, synth
    from gp_algos_amd.core import Context, EpClassifierState
    ctx = Context(local_rank)
    names = _class_names(L)
    prof_mask = (1 << L.GP_PROF_GEMM) | (1 << L.GP_PROF_SYRK) | (1 << L.GP_PROF_PANEL_UPD)

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        gdist.barrier()
        torch.cuda.synchronize()

    if args.workload == "c3":
        n = args.n if args.n != 8192 else 4096
        c3 = measure_c3(ctx, n, args.d, args.steps, args.warmup, rank, world, local_rank, backend, coll_dev, names)
        xin = c3.pop("_cross_check_inputs")
        if rank == 0:
            P, B = args.d + 2, c3["B"]
            print(json.dumps({"metric": "LML+gradient settings/sec at n=%d fp64, P=%d" % (n, P), "value": c3["settings_per_s"],
                              "unit": "settings/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": c3["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                              "dtype": "f64", "data": "synthetic",
                              "config": {"workload": c3["workload"], "B": B, "n": n, "timed_region": c3["timed_region"]},
                              "algorithmic_tflops_total": c3["algorithmic_tflops_total"],
                              "algorithmic_tflops_per_gpu": c3["algorithmic_tflops_per_gpu"],
                              "frac_of_fp64_mfma_peak_per_gpu": c3["frac_of_fp64_mfma_peak_per_gpu"],
                              "roofline": c3["roofline"], "per_rank_seconds": c3["per_rank_seconds"],
                              "lml_first": c3["lml_first"], "lml_last": c3["lml_last"], "all_finite": c3["all_finite"],
                              "lockstep_vs_single_setting_max_rel": c3["lockstep_vs_single_setting_max_rel"],
                              "c_abi_rccl_allgather_matches": c3["c_abi_rccl_allgather_matches"]}), flush=True)
        c3["_cross_check_inputs"] = xin
        c_abi_cross_check(ctx, c3, rank, world, local_rank, backend)      # after the line: cannot cost it
    elif args.workload == "c5":
        import ctypes as C
        n = args.n if args.n != 8192 else 32768
        m_total = args.m if args.m != 65536 else 1000000     # ONE request of 10^6 test points, split contiguously over the ranks
        p = synth.config_c5(n, args.d, 0)
        lo, hi = gdist.shard_range(m_total, rank, world)
        m = hi - lo
        i = (np.arange(m, dtype=np.uint64) + np.uint64(lo))[:, None]
        k = np.arange(args.d, dtype=np.uint64)[None, :]
        Xs = np.asfortranarray(-2.0 + 4.0 * synth.u(43, i * np.uint64(args.d) + k))
        lib = ctx._lib
        dX, dy, dXs = ctx.upload(p["X"]), ctx.upload(p["y"]), ctx.upload(Xs)
        dmean, dvar = ctx.dev_alloc(8 * max(m, 1)), ctx.dev_alloc(8 * max(m, 1))
        theta = L.f64(p["theta"])
        h, info = C.c_void_p(), C.c_int()
        t0 = time.perf_counter()
        ctx.check(lib.gp_fit_rbf_dev(ctx.h, dX, n, args.d, n, dy, L.dptr(theta), float("nan"), C.byref(h), C.byref(info)), info.value)
        ctx.sync()
        t_fit = time.perf_counter() - t0

        def step():   # this rank's slice of the request, then mean/variance assembled on every rank (2 m/G doubles per rank)
            ctx.check(lib.gp_predict_dev(h, dXs, m, m, dmean, dvar))
            local = np.stack([ctx.download(dmean, (m,)), ctx.download(dvar, (m,))], axis=1)
            return gdist.all_gather_rows(local, m_total, device=coll_dev)

        for _ in range(max(args.warmup, 1)):
            step()
        fence()
        ctx.profile(prof_mask)
        t_rank = time.perf_counter()
        for _ in range(args.steps):
            full = step()
        fence()
        t_rank = time.perf_counter() - t_rank
        ctx.profile(0)
        dt = gdist.max_over_ranks(t_rank, device=coll_dev)
        times = gdist.all_gather_rows(np.array([[t_rank]]), world, device=coll_dev) if world > 1 else np.array([[t_rank]])
        roof = _roofline_from_profile(ctx, L, (L.GP_PROF_GEMM, L.GP_PROF_SYRK), names, "c5")
        if roof:
            # posterior step i multiplies the m_b x 128 (i + 1) prefix of Vt by a 128 x 128 (i + 1) block row of Lw: averaged over the
            # n / 128 steps of a full batch (the PMC passes ran full batches of 131072 rows)
            mb = min(m, 131072)
            roof["algorithmic_bytes_per_launch"] = 8.0 * (mb * ((n + 128) / 2.0) + 128 * ((n + 128) / 2.0) + mb * 128)
        if rank == 0:
            var = full[:, 1]
            print(json.dumps({"metric": "posterior variances/sec at n=%d fp64 (L resident)" % n, "value": m_total * args.steps / dt,
                              "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, 1),
                              "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                              "dtype": "f64", "data": "synthetic",
                              "config": {"workload": "C5: n=%d d=%d fit once per GPU (first fit incl. allocation %.0f ms), ONE request of %d test "
                                                     "points split contiguously over the ranks (%d per GPU, batches of up to 131072 rows, Vt <= 32 GiB), "
                                                     "mean and variance gathered on every rank" % (n, args.d, t_fit * 1e3, m_total, m),
                                         "n": n, "m_total": m_total, "m_per_gpu": m,
                                         "timed_region": "per step: predict of this rank's slice (X* resident) + D2H of 2 m/G doubles + all_gather"},
                              "tflops_n2m_total": float(n) * n * m_total * args.steps / dt / 1e12,
                              "tflops_n2m_per_gpu": float(n) * n * m_total * args.steps / dt / 1e12 / world,
                              "roofline": roof, "per_rank_seconds": [float(t) for t in times[:, 0]],
                              "var_range": [float(var.min()), float(var.max())], "gathered_rows": int(full.shape[0])}), flush=True)
        lib.gp_model_destroy(h)
    else:
        n = args.n if args.n != 8192 else 4096
        sweeps = 50
        p = synth.config_c4(n, args.d)
        progress("c4: context up, building the Gram matrix")
        K = ctx.gram_rbf(p["X"], p["theta"], full=True)
        progress("c4: Gram matrix built (mirrored form), creating the EP state")
        ep = EpClassifierState(ctx, K, p["y"])
        progress("c4: EP state created, first sweep")
        ep.sweep(1)
        fence()
        progress("c4: warm-up sweep done, timed sweeps")
        t0 = time.perf_counter()
        for _ in range(args.steps):
            tau, nu = ep.sweep(sweeps)
            progress("c4: %d sweeps done" % sweeps)
        fence()
        progress("c4: fence after the timed sweeps passed")
        dt = gdist.max_over_ranks(time.perf_counter() - t0, device=coll_dev)
        # the roofline block's HIP events (two per launch of the profiled classes, on four streams) cost this latency-bound workload
        # ~5 %: they are taken on ONE extra step after the timed ones.  --no-roofline-events leaves that step out (the roofline block
        # is then null): a rocprofv3 --pmc pass only needs the kernels to run, and the WRITE_SIZE pass of this workload stopped
        # in exactly this step (profiles/r04_a_pmc_c4_WRITE_SIZE_probe.log)
        if not args.no_roofline_events:
            ctx.profile(prof_mask)
            for part in range(5):
                ep.sweep(sweeps // 5)
                progress("c4: %d profiled sweeps done" % ((part + 1) * (sweeps // 5)))
            fence()
            ctx.profile(0)
            progress("c4: profiled sweeps done")
        names_c4 = dict(names)
        names_c4[L.GP_PROF_SYRK] = ("gemm_nt_f64_kernel<1,*> / gemm_k128_kernel<1> (lower-trapezoid products of the refactorisation that runs under "
                                    "the site loop: K = 512 trailing updates, next covariance -= Vt Vt^T; launched on two side streams, so a "
                                    "launch shares the chip with up to three other streams and its duration includes that sharing)")
        roof = _roofline_from_profile(ctx, L, (L.GP_PROF_GEMM, L.GP_PROF_SYRK, L.GP_PROF_PANEL_UPD), names_c4, "c4")
        # the same EP run over a GRID of settings (MeshHyperParamsLogLikelihoodEvaluator.scala:26-40): 12 settings x 10 sweeps, in lockstep
        # (gp_ep_lml_rbf_batched's default) and one setting at a time; outside the timed region, every rank its own copy
        grid = None
        if not args.no_c3:
            th = p["theta"]
            thetas = np.stack([th * np.concatenate(([1.0 + 0.05 * b], np.ones(th.size - 2) * (1.0 + 0.03 * b), [1.0])) for b in range(12)])
            ctx.ep_lml_rbf_batched(p["X"], p["y"], thetas[:3], stop_eps=-1.0, max_sweeps=2)
            t1 = time.perf_counter()
            gl, _, _ = ctx.ep_lml_rbf_batched(p["X"], p["y"], thetas, stop_eps=-1.0, max_sweeps=10)
            t_lock = time.perf_counter() - t1
            os.environ["GPCORE_EP_LOCKSTEP"] = "0"
            t1 = time.perf_counter()
            gs, _, _ = ctx.ep_lml_rbf_batched(p["X"], p["y"], thetas, stop_eps=-1.0, max_sweeps=10)
            t_ser = time.perf_counter() - t1
            del os.environ["GPCORE_EP_LOCKSTEP"]
            grid = {"settings": 12, "sweeps_each": 10, "lockstep_sweeps_per_s": 120 / t_lock, "one_at_a_time_sweeps_per_s": 120 / t_ser,
                    "lockstep_executed_tflops": (8.0 / 3.0) * float(n) ** 3 * 120 / t_lock / 1e12,
                    # (the batch takes the next covariance in K = 1024 steps, a single run in K = 256: same sums, grouped differently)
                    "max_rel_diff_lml": float(np.max(np.abs(gl - gs) / np.abs(gs)))}
        if rank == 0:
            tf = (13.0 / 3.0) * float(n) ** 3 * sweeps * args.steps / dt / 1e12    # 4 1/3 n^3 per sweep
            tf_exec = (8.0 / 3.0) * float(n) ** 3 * sweeps * args.steps / dt / 1e12
            print(json.dumps({"metric": "EP sweeps/sec at n=%d fp64" % n, "value": world * sweeps * args.steps / dt, "unit": "sweeps/s",
                              "n_gpus": world, "steps": args.steps, "warmup": 1, "ms_per_step": dt / args.steps * 1e3,
                              "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                              "config": {"workload": "C4: GP binary classification via EP, n=%d, %d sweeps per step (replicas only: one EP run "
                                                     "does not shard)" % (n, sweeps), "n": n, "sweeps": sweeps},
                              "algorithmic_tflops": tf, "frac_of_fp64_mfma_peak": tf / PEAK_FP64_MFMA_TFLOPS,
                              "executed_tflops": tf_exec, "executed_frac_of_fp64_mfma_peak": tf_exec / PEAK_FP64_MFMA_TFLOPS,
                              "critical_path_note": "a sweep is bound by the serial site chain (ONE single-workgroup kernel per 128 sites: the link to "
                                                    "the block before as its prologue, then ~0.58 us per site, a chain of dependent fp64 "
                                                    "operations), not by a throughput roofline; all matrix work runs on three side streams under "
                                                    "it (profiles/: sweep summary, phase stamps of the chain kernel).  A grid of settings runs in "
                                                    "lockstep instead (gp_ep_lml_rbf_batched: profiles/*ep_mesh_perf*)",
                              "executed_flops_note": "the sweep executes 2 2/3 n^3 (trailing-only rank-128 updates n^3/3, Cholesky n^3/3, "
                                                     "V n^3, Sigma n^3), the 4 1/3 n^3 of SURVEY 8(d) counts full-square rank-1 updates",
                              "roofline": roof, "ep_grid": grid,
                              "ep_lml_strict": ep.lml(True), "ep_lml_corrected": ep.lml(False),
                              "tau_range": [float(tau.min()), float(tau.max())]}), flush=True)
        ep.close()
    ctx.close()
    gdist.barrier()


def _class_names(L):
    return {L.GP_PROF_GEMM: "gemm_nt_f64_kernel<0,*> (general product: T = L^-T updates / rank-128 EP updates / posterior steps)",
            L.GP_PROF_SYRK: "gemm_nt_f64_kernel<1,*> (lower-trapezoid product: Cholesky K=512 trailing update, T T^T, Sigma = K - Vt Vt^T)",
            L.GP_PROF_PANEL_UPD: "gemm_nt_f64_kernel<1,1> (K = 128 in-panel update)"}


def measure_c3(ctx, n, d, steps, warmup, rank, world, local_rank, backend, coll_dev, names):
    """BASELINE config C3, the north-star's scaling workload: LML + gradient over the 64 settings of the 4 x 4 x 4 grid at
    n = 4096, settings sharded b -> rank b // ceil(B / world) (8 per GPU at 8 GPUs), every rank holds X and y, results
    assembled on every rank by ONE all_gather per step (gp_algos_amd/dist.py lml_grad_sharded; strong scaling).  Timed like the
    headline: barrier + device sync on both sides, max over ranks.  Outside the timed region the same sharded evaluation runs
    through the C-ABI's own RCCL entry point (gp_dist_lml_grad_batched -- what a Scala host calls) and is compared with it.
    Returns the figures on every rank."""
    import torch
    from gp_algos_amd import _lib as L, dist as gdist, synth
    P = d + 2
    p = synth.config_c3(n, d)
    B = p["thetas"].shape[0]
    lo, hi = gdist.shard_range(B, rank, world)

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        gdist.barrier()
        torch.cuda.synchronize()

    def evaluate(th):                                    # this rank's settings on this rank's GPU
        lml_, grad_, _ = ctx.lml_grad_batched(p["X"], p["y"], th)
        return lml_, grad_

    def step():                                          # results by ONE all_gather (after the status exchange)
        return gdist.lml_grad_sharded(evaluate, p["thetas"], device=coll_dev)

    for _ in range(warmup):
        step()                                           # same shapes as the timed steps: workspaces are allocated here
    fence()
    prof_mask = (1 << L.GP_PROF_GEMM) | (1 << L.GP_PROF_SYRK) | (1 << L.GP_PROF_PANEL_UPD)
    ctx.profile(prof_mask)
    t_rank = time.perf_counter()
    for _ in range(steps):
        lml, grad = step()
    fence()
    t_rank = time.perf_counter() - t_rank
    ctx.profile(0)
    dt = gdist.max_over_ranks(t_rank, device=coll_dev)
    times = gdist.all_gather_rows(np.array([[t_rank]]), world, device=coll_dev) if world > 1 else np.array([[t_rank]])
    roof = _roofline_from_profile(ctx, L, (L.GP_PROF_GEMM, L.GP_PROF_SYRK, L.GP_PROF_PANEL_UPD), names, "c3")
    # spot check outside the timed region: one setting alone (count = 1 forms of every step) against its lockstep result
    b0 = lo if hi > lo else 0
    one, gone, _ = ctx.lml_grad_batched(p["X"], p["y"], p["thetas"][b0:b0 + 1])
    chk = float(max(abs(one[0] - lml[b0]) / abs(one[0]), np.max(np.abs(gone[0] - grad[b0])) / np.max(np.abs(gone[0]))))
    # n^3/3 (potrf) + 2 n^3/3 (K^-1 from L) + 2 n^2 (alpha) + P 2 n^2 (fused traces) per setting
    flops = float(n) ** 3 + (2.0 + 2.0 * P) * n * n
    tf = flops * B * steps / dt / 1e12
    return {"settings_per_s": B * steps / dt, "unit": "settings/s", "scaling": "strong", "n_gpus": world, "steps": steps,
            "warmup": warmup, "ms_per_step": dt / steps * 1e3, "B": B, "n": n, "settings_per_gpu": hi - lo,
            "workload": "C3: log-marginal-likelihood + gradient over %d hyper-parameter settings, n=%d d=%d, settings sharded %d per GPU "
                        "(b -> rank b // %d), every rank holds X and y, results assembled on every rank by one all_gather of "
                        "(B/G) x (1+P) doubles per step" % (B, n, d, -(-B // world), -(-B // world)),
            "timed_region": "per step: upload of X, y (%d KB) + all settings of this rank + status exchange + the all_gather" % ((n * d + n) * 8 // 1024),
            "algorithmic_tflops_total": tf, "algorithmic_tflops_per_gpu": tf / world,
            "frac_of_fp64_mfma_peak_per_gpu": tf / world / PEAK_FP64_MFMA_TFLOPS, "roofline": roof,
            "per_rank_seconds": [float(t) for t in times[:, 0]], "lml_first": float(lml[0]), "lml_last": float(lml[-1]),
            "all_finite": bool(np.all(np.isfinite(lml)) and np.all(np.isfinite(grad))),
            "lockstep_vs_single_setting_max_rel": chk,
            "c_abi_rccl_allgather_matches": ("checked AFTER this line is printed (c_abi_cross_check); result as one JSON line on stderr"
                                             if world > 1 and backend == "nccl" else None),
            "_cross_check_inputs": (p, lml, grad)}


def c_abi_cross_check(ctx, c3, rank, world, local_rank, backend):
    """The C3 evaluation once more through the C-ABI's own RCCL path (gp_dist_unique_id -> gp_dist_init -> gp_dist_lml_grad_batched:
    what a Scala host calls), compared bit for bit with the torch.distributed result.  Runs AFTER rank 0 has printed and flushed the
    benchmark line, so nothing that goes wrong here can cost the line; its result goes to stderr as one JSON
    line.  A failure on ONE rank before the communicator exists must not strand the others inside ncclCommInitRank: every local
    step is followed by a torch all_reduce(MIN) of "this rank is fine", and all ranks skip together when any rank is not."""
    import torch
    import torch.distributed as tdist
    from gp_algos_amd import _lib as L
    if not (world > 1 and backend == "nccl" and c3 is not None):
        return None
    p, lml, grad = c3["_cross_check_inputs"]
    lib = ctx._lib
    dev = torch.device("cuda", local_rank)

    def everybody(ok):
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        tdist.all_reduce(t, op=tdist.ReduceOp.MIN)
        return bool(t.item())

    result, h = None, C.c_void_p()
    try:
        buf = C.create_string_buffer(L.GP_DIST_ID_BYTES)
        st = lib.gp_dist_unique_id(ctx.h, buf)          # every rank: resolves RCCL here (only rank 0's id is used)
        if not everybody(st == 0):
            result = "skipped: RCCL could not be resolved on some rank (this rank: status %d %s)" % (st, lib.gp_last_error(ctx.h).decode() if st else "")
        else:
            ident = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).to(dev)
            tdist.broadcast(ident, src=0)               # a plain byte tensor: nothing is pickled
            raw = bytes(ident.cpu().numpy().tobytes())
            st = lib.gp_dist_init(ctx.h, C.create_string_buffer(raw, L.GP_DIST_ID_BYTES), rank, world, C.byref(h))
            if not everybody(st == 0):
                result = "skipped: gp_dist_init failed on some rank (this rank: status %d %s)" % (st, lib.gp_last_error(ctx.h).decode() if st else "")
            else:
                from gp_algos_amd.core import DistGroup
                grp = DistGroup.__new__(DistGroup)
                grp.ctx, grp.rank, grp.world, grp.h = ctx, rank, world, h
                h = C.c_void_p()
                l2, g2, _ = grp.lml_grad_batched(p["X"], p["y"], p["thetas"])
                result = bool(np.array_equal(l2, lml) and np.array_equal(g2, grad))
                grp.close()
    except Exception as e:                              # reported, never fatal
        result = "failed: %s" % e
    if h.value:
        lib.gp_dist_destroy(h)
    if rank == 0:
        print(json.dumps({"c_abi_rccl_allgather_matches": result, "n_gpus": world,
                          "librccl_mapped": sorted({ln.split()[-1] for ln in open("/proc/self/maps") if "librccl" in ln})}),
              file=sys.stderr, flush=True)
    return result


def self_launch(gpus, argv, dry=False):
    """`bench.py --gpus N` with no launcher around it: start the N ranks as child processes of THIS process -- which has not
    touched the GPU and never will (no torch import, no HIP call, no libgpcore.so) -- with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR=127.0.0.1 / MASTER_PORT in their environment, wait for them and return the worst exit status.  Rank 0 inherits
    stdout (it prints the one JSON line); the other ranks' stdout goes to stderr.  If a rank dies, the rest are given 30 s to
    follow (their collectives fail or time out) and are then terminated by PID."""
    import socket
    import subprocess
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    procs = []
    for r in range(gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(gpus), LOCAL_WORLD_SIZE=str(gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GPCORE_BENCH_SELF_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if (r == 0 or dry) else sys.stderr))
    rc, deadline = 0, None
    while procs:
        for pr in list(procs):
            code = pr.poll()
            if code is None:
                continue
            procs.remove(pr)
            if code != 0:
                rc = rc or (code if code > 0 else 128 - code)
                deadline = deadline or time.time() + 30.0
        if procs and deadline and time.time() > deadline:
            for pr in procs:
                pr.terminate()
            deadline = time.time() + 10.0
        time.sleep(0.05)
    return rc


PMC_SUMMARY = PMC_TAG + "_pmc_c2_summary.json"
DOMINANT_KERNEL = "gemm_fused_kernel<0,0,1,8>"


def cpu_opt_baseline(p, sample_pts=4096):
    """Optimised-CPU comparator (numpy/scipy on multithreaded LAPACK, all host cores): vectorised Gram, cho_factor,
    cho_solve, solve_triangular on a bounded sample of the test points.  This is the figure the north-star's
    ">= 10x the CPU wall-clock for full fit+predict" is judged against; the 1-core oracle above is the parity checker."""
    import scipy.linalg as sla
    cores = host_cores()
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=cores)
    except Exception:
        limiter = None
    X, y, theta = p["X"], p["y"], p["theta"]
    n = X.shape[0]
    sf2, sn2 = theta[0] ** 2, theta[-1] ** 2
    Z = X / theta[1:-1]
    t0 = time.perf_counter()
    sq = (Z * Z).sum(axis=1)
    K = sq[:, None] + sq[None, :] - 2.0 * (Z @ Z.T)
    np.maximum(K, 0.0, out=K)
    np.exp(-0.5 * K, out=K)
    K *= sf2
    K[np.diag_indices(n)] = sf2 + sn2
    c = sla.cho_factor(K, lower=True, overwrite_a=True, check_finite=False)
    alpha = sla.cho_solve(c, y, check_finite=False)
    t_fit = time.perf_counter() - t0
    k = min(sample_pts, p["Xs"].shape[0])
    Zs = p["Xs"][:k] / theta[1:-1]
    t0 = time.perf_counter()
    Ks = (Zs * Zs).sum(axis=1)[:, None] + sq[None, :] - 2.0 * (Zs @ Z.T)
    np.maximum(Ks, 0.0, out=Ks)
    np.exp(-0.5 * Ks, out=Ks)
    Ks *= sf2
    mean = Ks @ alpha
    V = sla.solve_triangular(c[0], Ks.T, lower=True, check_finite=False, overwrite_b=True)
    var = (sf2 + sn2) - np.einsum("ij,ij->j", V, V)
    t_pred = time.perf_counter() - t0
    if limiter is not None:
        limiter.restore_original_limits()
    return dict(kind="numpy/scipy LAPACK, BLAS threads = the cores this process may use", cores=cores, visible_cpus=os.cpu_count(),
                fit_s=t_fit, predict_points_per_s=k / t_pred,
                sample="fit at full n=%d; predict on %d of the test points" % (n, k)), mean, var, k


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--train-n", dest="n", type=int, default=8192)
    ap.add_argument("--dim", dest="d", type=int, default=8)
    ap.add_argument("--test-points", dest="m", type=int, default=65536)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c4", "c5"],
                    help="c2 (default, the BASELINE.json metric): fit + posterior; c3: batched LML+gradient over 64 settings; "
                         "c4: EP classification sweeps; c5: n=32768 fit once, then 10^6/8 test-point variances per GPU per step")
    ap.add_argument("--dry-launch", action="store_true",
                    help="ranks print their RANK / LOCAL_RANK / WORLD_SIZE as one JSON line and exit without touching the GPU "
                         "(the launch path of --gpus N, testable on a host without a GPU)")
    ap.add_argument("--cpu-fit-full", action="store_true", help="cpu_baseline: time the oracle's fit at the full n (about a minute)")
    ap.add_argument("--no-c3", action="store_true", help="leave the c3_sharded block out of the c2 line")
    ap.add_argument("--no-roofline-events", action="store_true",
                    help="c4: skip the extra step that records HIP events around every launch of the profiled classes (for rocprofv3 --pmc passes)")
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher around us: become one.  Nothing above this line imports torch or loads libgpcore.so.
        raise SystemExit(self_launch(args.gpus, sys.argv[1:], dry=args.dry_launch))
    if args.dry_launch:
        if os.environ.get("GPCORE_BENCH_DRY_FAIL_RANK") == os.environ.get("RANK", "0"):
            raise SystemExit(3)                  # test hook: one rank dies, the launcher must report it
        print(json.dumps({"dry_launch": True, "rank": int(os.environ.get("RANK", "0")), "local_rank": int(os.environ.get("LOCAL_RANK", "0")),
                          "world": int(os.environ.get("WORLD_SIZE", "1")), "master": "%s:%s" % (os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT")),
                          "self_launched": os.environ.get("GPCORE_BENCH_SELF_LAUNCHED") == "1"}), flush=True)
        return
    if "WORLD_SIZE" in os.environ and args.gpus != int(os.environ["WORLD_SIZE"]):
        print("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks; the launcher wins" % (args.gpus, os.environ["WORLD_SIZE"]), file=sys.stderr)
    if args.workload != "c2":
        return run_secondary(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
    # Rehearsal knobs for a 1-GPU box only (never set by the driver): run several ranks on one device over gloo.
    if os.environ.get("GPCORE_BENCH_DEVICE") is not None:
        local_rank = int(os.environ["GPCORE_BENCH_DEVICE"])
    backend = os.environ.get("GPCORE_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import __graft_entry__ as entry
    entry.build()
    from gp_algos_amd import _lib as L, synth
    from gp_algos_amd.core import Context

    n, d, m = args.n, args.d, args.m
    p = synth.config_c2(n, d, 0)
    # every rank predicts its own slice of the test stream (seed 13, offset by rank)
    i = (np.arange(m, dtype=np.uint64) + np.uint64(rank * m))[:, None]
    k = np.arange(d, dtype=np.uint64)[None, :]
    p["Xs"] = np.asfortranarray(-2.0 + 4.0 * synth.u(13, i * np.uint64(d) + k))

    ctx = Context(local_rank)
    lib = ctx._lib
    probe = ctx.probe_mfma_f64()
    dX, dy, dXs = ctx.upload(p["X"]), ctx.upload(p["y"]), ctx.upload(p["Xs"])
    gathered = None
    if world > 1 and backend == "nccl":
        # N > 1: the posterior of the N*m test points is ASSEMBLED on every rank -- this rank's (mean | variance) slice lives in
        # a torch tensor the library writes through its raw pointer, and one all_gather per step (RCCL over xGMI, 2 m doubles
        # = 1 MB per rank) follows the predict inside the timed region
        dout_t = torch.empty(2 * m, dtype=torch.float64, device="cuda")
        gathered = torch.empty(world * 2 * m, dtype=torch.float64, device="cuda")
        dmean, dvar = C.c_void_p(dout_t.data_ptr()), C.c_void_p(dout_t.data_ptr() + 8 * m)
    else:
        dmean, dvar = ctx.dev_alloc(8 * m), ctx.dev_alloc(8 * m)
    theta = L.f64(p["theta"])
    nan = float("nan")
    h, info = C.c_void_p(), C.c_int()
    ctx.check(lib.gp_fit_rbf_dev(ctx.h, dX, n, d, n, dy, L.dptr(theta), nan, C.byref(h), C.byref(info)), info.value)

    def step():
        ctx.check(lib.gp_model_refit_dev(h, L.dptr(theta), nan))
        ctx.check(lib.gp_predict_dev(h, dXs, m, m, dmean, dvar))
        if gathered is not None:
            ctx.sync()                                        # the library runs on its own stream; torch's collective on torch's
            dist.all_gather_into_tensor(gathered, dout_t)

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    prof_mask = (1 << L.GP_PROF_GEMM) | (1 << L.GP_PROF_SYRK) | (1 << L.GP_PROF_GRAM)
    ctx.profile(prof_mask)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    ctx.profile(0)
    g_k, g_ms, g_work = ctx.profile_read(L.GP_PROF_GEMM)
    s_k, s_ms, s_work = ctx.profile_read(L.GP_PROF_SYRK)
    r_k, r_ms, r_work = ctx.profile_read(L.GP_PROF_GRAM)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # stage breakdown (untimed, rank 0 only needs it)
    ctx.sync()
    t1 = time.perf_counter()
    ctx.check(lib.gp_model_refit_dev(h, L.dptr(theta), nan))
    ctx.sync()
    t_fit = time.perf_counter() - t1
    t1 = time.perf_counter()
    ctx.check(lib.gp_predict_dev(h, dXs, m, m, dmean, dvar))
    ctx.sync()
    t_pred = time.perf_counter() - t1
    ctx.check(lib.gp_model_status(h, C.byref(info)), info.value)
    # End to end as SURVEY.md 8(d) "GPU timing" asks: one step INCLUDING the transfers -- H2D of X, y and X* (pageable host arrays,
    # as a JNI caller hands them over), refit, predict, D2H of mean and variance.  Reported beside `value`, never as `value`.
    hX, hy, hXs = np.asfortranarray(p["X"]), np.ascontiguousarray(p["y"]), np.asfortranarray(p["Xs"])
    hmean, hvar = np.empty(m), np.empty(m)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    e2e = []
    for _ in range(3):
        ctx.sync()
        t1 = time.perf_counter()
        ctx.check(lib.gp_dev_upload(ctx.h, dX, vp(hX), hX.nbytes))
        ctx.check(lib.gp_dev_upload(ctx.h, dy, vp(hy), hy.nbytes))
        ctx.check(lib.gp_dev_upload(ctx.h, dXs, vp(hXs), hXs.nbytes))
        ctx.check(lib.gp_model_refit_dev(h, L.dptr(theta), nan))
        ctx.check(lib.gp_predict_dev(h, dXs, m, m, dmean, dvar))
        ctx.check(lib.gp_dev_download(ctx.h, vp(hmean), dmean, hmean.nbytes))
        ctx.check(lib.gp_dev_download(ctx.h, vp(hvar), dvar, hvar.nbytes))
        ctx.sync()
        e2e.append(time.perf_counter() - t1)
    t_e2e = sorted(e2e)[1]
    # The trailing-update kernels by themselves: one refit with the look-ahead off (overlapping launches would each be timed
    # with the other running beside them), HIP events around every launch of the two classes
    ctx.check(lib.gp_ctx_set_lookahead(ctx.h, 0))
    ctx.profile((1 << L.GP_PROF_SYRK) | (1 << L.GP_PROF_PANEL_UPD))
    t1 = time.perf_counter()
    ctx.check(lib.gp_model_refit_dev(h, L.dptr(theta), nan))
    ctx.sync()
    t_fit_serial = time.perf_counter() - t1
    ctx.profile(0)
    ctx.check(lib.gp_ctx_set_lookahead(ctx.h, -1))
    o_k, o_ms, o_work = ctx.profile_read(L.GP_PROF_SYRK)
    p_k, p_ms, p_work = ctx.profile_read(L.GP_PROF_PANEL_UPD)

    # The north-star's scaling workload in the same line: C3 sharded over the ranks (every rank takes part: collectives inside)
    c3, c3_inputs = None, None
    if not args.no_c3:
        c3 = measure_c3(ctx, 4096, d, max(2, min(args.steps, 5)), 1, rank, world, local_rank, backend,
                        "cuda" if backend == "nccl" else "cpu", _class_names(L))
        c3_inputs = c3.pop("_cross_check_inputs")

    out = None
    if rank == 0:
        # HBM traffic per launch of the dominant kernel: PMC counters cannot be read from inside the process, so the
        # figure comes from the committed rocprofv3 --pmc passes of this same command (profiles/, FETCH_SIZE doubled
        # per the gfx950 correction); only reported for the configuration those passes ran.
        traffic, traffic_note = None, None
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", PMC_SUMMARY)))
            if pm.get("kernel_sources_sha256") != kernel_sources_sha256():
                traffic_note = "profiles/%s was collected on other kernel sources than this tree's: not reported" % PMC_SUMMARY
            elif (n, d, m) == (8192, 8, 65536):
                traffic = pm["kernels"][DOMINANT_KERNEL]["hbm_bytes_per_launch_corrected"]
        except Exception:
            traffic = None
        ms_per_step = dt / args.steps * 1e3
        value = world * m * args.steps / dt
        gemm_tflops = g_work / (g_ms * 1e-3) / 1e12 if g_ms > 0 else 0.0
        syrk_tflops = s_work / (s_ms * 1e-3) / 1e12 if s_ms > 0 else 0.0
        out = {
            "metric": "GP posterior (mu,var) points/sec at n=%d fp64 (fit + predict per step)" % n,
            "value": value, "unit": "points/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C2: GP regression, ARD-RBF, n=%d d=%d fp64, Gram + Cholesky + posterior mean/variance at m=%d "
                                   "test points per GPU" % (n, d, m), "n": n, "d": d, "m_per_gpu": m,
                       "parallelism": ("test points sharded %d-way (m per GPU), model refit per rank; the posterior of all %d points is "
                                       "assembled on every rank by one all_gather of 2 m doubles per rank per step (RCCL)" % (world, world * m))
                       if gathered is not None else "single GPU" if world == 1 else
                       "test points sharded %d-way, model refit per rank, results left on the rank (rehearsal backend)" % world,
                       "timed_region": "X, y, X* resident in HBM before the timed steps; per step: Gram + Cholesky + alpha/LML refit, then mean "
                                       "and variance of all m points, results left in HBM (H2D of X* = m*d*8 B and D2H of 2m doubles are outside "
                                       "the timed region: 5 MB per step, DESIGN.md section 5)"},
            "roofline": {"kernel": DOMINANT_KERNEL + " (posterior step Vt_i = Vt[:, :128(i+1)] Lw_i^T: update and panel solve of block "
                                   "column i in one product on 256x128 tiles, one 8-wave workgroup (64x64 per wave) per CU, row reductions in the "
                                   "epilogue; v_mfma_f64_16x16x4_f64)",
                         "bound": "mfma", "achieved": gemm_tflops, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": gemm_tflops / PEAK_FP64_MFMA_TFLOPS, "traffic": traffic, "traffic_note": traffic_note,
                         "traffic_unit": "HBM bytes per launch (rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/%s)" % PMC_SUMMARY,
                         # average over the n/128 launches, K_i = 128 (i + 1): A read m K, Lw block row read 128 K, C written m 128
                         "algorithmic_bytes_per_launch": 8.0 * (m * ((n + 128) / 2.0) + 128 * ((n + 128) / 2.0) + m * 128),
                         "launches": g_k, "avg_launch_us": g_ms / max(g_k, 1) * 1e3,
                         "flops_per_launch_avg": g_work / max(g_k, 1)},
            "cholesky": _cholesky_block(n, t_fit, t_fit_serial, o_k, o_ms, o_work, p_k, p_ms, p_work),
            "gram": {"GBps": r_work / (r_ms * 1e-3) / 1e9 if r_ms > 0 else 0.0, "frac_of_hbm_peak": (r_work / (r_ms * 1e-3) / 1e9) / PEAK_HBM_GBS if r_ms > 0 else 0.0},
            "predict_only_points_per_s": m / t_pred, "predict_ms": t_pred * 1e3,
            "e2e_ms_per_step": t_e2e * 1e3, "e2e_points_per_s": m / t_e2e,
            "e2e_note": "one step INCLUDING transfers (SURVEY.md 8(d) GPU timing): H2D of X, y, X* (%.1f MB, pageable host memory) + refit + "
                        "predict + D2H of mean and variance (%.1f MB); median of 3; rank 0's GPU only, not part of `value`"
                        % ((hX.nbytes + hy.nbytes + hXs.nbytes) / 1e6, 2 * hmean.nbytes / 1e6),
            "mfma_f64_probe_tflops": probe,
            "c3_sharded": c3,
        }
    # CPU baseline on rank 0 at N=1 only
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        Lh = np.zeros((n, n), order="F")
        ctx.check(lib.gp_model_get(h, L.GP_GET_L, L.dptr(Lh), n))
        ah = np.zeros(n)
        ctx.check(lib.gp_model_get(h, L.GP_GET_ALPHA, L.dptr(ah), n))
        base, cmean, cvar, kk = cpu_baseline(p, Lh, ah, m, fit_full_size=args.cpu_fit_full)
        gmean = ctx.download(dmean, (m,))[:kk]
        gvar = ctx.download(dvar, (m,))[:kk]
        base["max_abs_dmean_vs_gpu"] = float(np.max(np.abs(gmean - cmean)))
        base["max_abs_dvar_vs_gpu"] = float(np.max(np.abs(gvar - cvar)))
        out["cpu_baseline"] = base
        opt, omean, ovar, ok = cpu_opt_baseline(p)
        gm, gv = ctx.download(dmean, (m,))[:ok], ctx.download(dvar, (m,))[:ok]
        opt["max_abs_dmean_vs_gpu"] = float(np.max(np.abs(gm - omean)))
        opt["max_abs_dvar_vs_gpu"] = float(np.max(np.abs(gv - ovar)))
        opt["fit_plus_predict_s_extrapolated_to_m"] = opt["fit_s"] + m / opt["predict_points_per_s"]
        opt["gpu_step_speedup_vs_cpu_opt"] = opt["fit_plus_predict_s_extrapolated_to_m"] / (ms_per_step * 1e-3)
        out["cpu_opt_baseline"] = opt
    if rank == 0:
        print(json.dumps(out), flush=True)
    if c3 is not None and c3_inputs is not None:
        # only now, with the line out: the C-ABI's own RCCL path against the torch.distributed result (stderr)
        c3["_cross_check_inputs"] = c3_inputs
        c_abi_cross_check(ctx, c3, rank, world, local_rank, backend)
    lib.gp_model_destroy(h)
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
