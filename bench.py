#!/usr/bin/env python3
"""bench.py -- headline measurement: vectors/s of Pq::quantize_batch (d=300, M=15, K=256) on
MI355X, inputs resident in HBM, through the C ABI of libpqhip.so.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch: rows_per_gpu synthetic fp32 vectors already
in HBM -> u8 codes in HBM.  With N > 1 every rank owns an independent row shard (the path has no
exchange step: no collective on the data path, "weak" scaling); the only collectives are the
barrier and the max-reduction of the elapsed time.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

D, M, K = 300, 15, 256            # BASELINE.json metric: d=300, M=15, K=256
DSUB = D // M
FLOP_PER_VEC = 2 * K * D          # distance GEMM only: 153,600
BYTES_PER_VEC = 4 * D + M         # algorithmic HBM bytes per vector: 1,215


def set_shape(d, m, k=256):
    """non-headline shapes (e.g. BASELINE configs[4]: d=768, M=48) for exploration runs"""
    global D, M, K, DSUB, FLOP_PER_VEC, BYTES_PER_VEC
    D, M, K = d, m, k
    DSUB = D // M
    FLOP_PER_VEC = 2 * K * D
    BYTES_PER_VEC = 4 * D + M

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: FP32 matrix peak (dense)
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec peak

WORKLOADS = {
    "encode": "Pq::quantize_batch {rows} x d=300 fp32 per GPU, M=15, K=256 (BASELINE configs[1]: 10M on 1 MI355X)",
    "opq_encode": "Opq rotate+encode {rows} x d=300 per GPU, M=15, K=256 (BASELINE configs[2])",
    "reconstruct": "Pq::reconstruct_batch {rows} u8 codes -> d=300 fp32 per GPU (BASELINE configs[3])",
    "lookup": "embedding lookup: {rows} random rows of a resident 10M x 15 u8 code matrix -> select + reconstruct + per-row rescale, d=300 fp32 (SURVEY 8f rank 2)",
    "opq_train": "device part of Opq::train_iteration (rotate, k-means update, quantize->reconstruct, X^T.R), {rows} x d=300 per GPU, M=15, K=256",
    "kmeans": "kmeans_iteration on all 15 subquantizers (training step; SURVEY 8f rank 1), {rows} x d=300 per GPU, K=256",
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows", type=int, default=10_000_000, help="rows per GPU")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="encode")
    ap.add_argument("--d", type=int, default=300)
    ap.add_argument("--m", type=int, default=15)
    ap.add_argument("--k", type=int, default=256, help="centroids per subquantizer (u8 codes up to 256, 32-bit codes beyond)")
    ap.add_argument("--variant", type=int, default=0, help="0 auto, 1 anchor kernel, 2 MFMA kernel")
    ap.add_argument("--cpu-rows", type=int, default=2_000_000, help="rows of the CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend for the barrier / max-reduction (gloo: rehearsals)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal on a one-GPU box: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--dry-run", action="store_true",
                    help="exercise sharding/reduction/printing only (CPU, gloo); no GPU work")
    return ap.parse_args()


def shard_ranges(world, rows):
    """weak scaling: rank r owns global rows [r*rows, (r+1)*rows)."""
    return [[r * rows, (r + 1) * rows] for r in range(world)]


def load_pmc_traffic(workload, rows):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/), if they were
    collected for exactly this workload size; else None."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        rec = json.load(open(p)).get(workload)
        if rec and int(rec["rows"]) == rows:
            return rec["hbm_bytes_per_launch"]
    except Exception:
        pass
    return None


def main():
    args = parse()
    if (args.d, args.m, args.k) != (300, 15, 256):
        set_shape(args.d, args.m, args.k)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_gpu = not args.dry_run
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=args.backend if use_gpu else "gloo", rank=rank, world_size=world)
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (args.gpus, world))
    rows = args.rows
    shards = shard_ranges(world, rows)

    def barrier():
        if world > 1:
            dist.barrier()

    kernel_ms = None
    extra = {}
    q0 = None
    if use_gpu:
        import numpy as np
        import reductive_amd
        import synth
        if args.single_device:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        reductive_amd.lib()                       # fails loudly if the HIP library is missing
        from reductive_amd.pq import _Ctx
        ctx = _Ctx(devices=[local_rank])          # one process per GPU
        q = synth.normalish(43, (M, K, DSUB))     # same codebook on every rank (replicated)
        P = synth.orthonormal(44, D) if args.workload == "opq_encode" else None
        pq = reductive_amd.Pq(P, q, ctx=ctx)
        if args.variant:
            pq.set_encode_variant(args.variant)
        g = torch.Generator(device=dev).manual_seed(42 + rank)
        if args.workload == "reconstruct":
            src = torch.randint(0, K, (rows, M), device=dev, dtype=torch.uint8, generator=g)
            dst = torch.empty((rows, D), device=dev, dtype=torch.float32)

            def step():
                pq.reconstruct_batch_device(src, out=dst)
        elif args.workload == "lookup":
            n_codes = 10_000_000
            src = torch.randint(0, K, (n_codes, M), device=dev, dtype=torch.uint8, generator=g)
            sel = torch.randint(0, n_codes, (rows,), device=dev, dtype=torch.int64, generator=g)
            scl = torch.rand((n_codes,), device=dev, dtype=torch.float32, generator=g) + 0.5
            dst = torch.empty((rows, D), device=dev, dtype=torch.float32)

            def step():
                pq.reconstruct_rows_device(src, sel, scales=scl, out=dst)
        else:
            src = torch.empty((rows, D), device=dev, dtype=torch.float32)
            for r0 in range(0, rows, 1 << 20):    # N(0,1) like benches/pq.rs:9, generated in HBM
                src[r0:r0 + (1 << 20)].normal_(generator=g)
            dst = torch.empty((rows, M), device=dev, dtype=torch.uint8 if K <= 256 else torch.int32)

            def step():
                pq.quantize_batch_device(src, out=dst)
        if args.workload == "opq_train":
            from reductive_amd.pq import opq_train_step
            P0 = synth.orthonormal(44, D)
            pick = torch.arange(K, device=dev) * (rows // K)
            q0 = np.stack([src[(pick + 7 * m) % rows, m * DSUB:(m + 1) * DSUB].cpu().numpy() for m in range(M)])
            state = {"q": q0}

            def step():
                state["q"], state["cross"] = opq_train_step(state["q"], P0, src, ctx=ctx)
        if args.workload == "kmeans":
            # one step = one kmeans_iteration (assign + update) of all M subquantizers over the
            # resident instances; the K timed steps are ONE library call with n_iterations = K,
            # exactly how pq.rs:176 drives it (centroids carried from step to step)
            from reductive_amd.pq import kmeans_iterations
            # initial centroids = K distinct instances per subquantizer (RandomInstanceCentroids, pq.rs:166-172)
            pick = torch.arange(K, device=dev) * (rows // K)
            q0 = np.stack([src[(pick + 7 * m) % rows, m * DSUB:(m + 1) * DSUB].cpu().numpy() for m in range(M)])

            def run_steps(k):
                if k > 0:
                    kmeans_iterations(q0, src, n_iterations=k, want_loss=False, ctx=ctx)
            run_steps(args.warmup)
            torch.cuda.synchronize()
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run_steps(args.steps)
            torch.cuda.synchronize()
            barrier()
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - t0
            kernel_ms = 1e3 * elapsed / args.steps
            extra["encode_kernel"] = "k_encode_mfma_lds3<vec4>"
            step = None
        for _ in range(args.warmup if step else 0):
            step()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
               for _ in range(args.steps)]
        if step:
            torch.cuda.synchronize()
            barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for a, b in evs:
                a.record()                        # same stream the kernels are launched on
                step()
                b.record()
            torch.cuda.synchronize()
            barrier()
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - t0
            kernel_ms = sum(a.elapsed_time(b) for a, b in evs) / len(evs)
            extra["encode_kernel"] = pq.last_encode_kernel() if args.workload not in ("reconstruct", "lookup") else "k_reconstruct"
    else:
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            time.sleep(0.001)
        barrier()
        elapsed = time.perf_counter() - t0

    tmax = torch.tensor([elapsed], dtype=torch.float64,
                        device="cuda" if (use_gpu and world > 1 and args.backend == "nccl") else "cpu")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())

    if rank == 0:
        total_rows = world * rows
        value = total_rows * args.steps / elapsed
        shape = "(d=%d, M=%d, K=%d)" % (D, M, K)
        names = {"encode": "vectors/sec PQ encode " + shape,
                 "opq_encode": "vectors/sec OPQ rotate+encode " + shape,
                 "reconstruct": "vectors/sec PQ reconstruct " + shape,
                 "lookup": "vectors/sec select+reconstruct+rescale lookup " + shape,
                 "opq_train": "vectors/sec per OPQ training iteration (device part) " + shape,
                 "kmeans": "vectors/sec per k-means iteration, all subquantizers " + shape}
        rec = {
            "metric": names[args.workload], "value": value, "unit": "vectors/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": WORKLOADS[args.workload].format(rows=rows) if (D, M, K) == (300, 15, 256) else
                       "%s, non-headline shape d=%d M=%d K=%d" % (args.workload, D, M, K), "rows_per_gpu": rows,
                       "rows_total": total_rows, "d": D, "M": M, "K": K, "shards": shards,
                       "placement": "inputs and outputs resident in HBM; C ABI device entry point"},
        }
        rec.update(extra)
        if use_gpu:
            sec = kernel_ms * 1e-3
            traffic = load_pmc_traffic(args.workload, rows)
            if args.workload in ("reconstruct", "lookup"):
                if args.workload == "lookup":
                    global BYTES_PER_VEC
                    BYTES_PER_VEC = 4 * D + M + 8 + 4     # output row + code row + row index + scale
                ach = BYTES_PER_VEC * rows / sec / 1e9
                rec["roofline"] = {"bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                   "frac": ach / PEAK_HBM_GBS, "traffic": traffic,
                                   "kernel": "k_reconstruct" + ("<.., SEL>" if args.workload == "lookup" else ""), "avg_launch_ms": kernel_ms,
                                   "algorithmic_bytes_per_vector": BYTES_PER_VEC}
            else:
                flop = FLOP_PER_VEC + (2 * D * D if args.workload == "opq_encode" else 0)
                if args.workload == "opq_train":   # rotation + 2 assignments + cross product
                    flop = 2 * D * D + 2 * FLOP_PER_VEC + 2 * D * D
                ach = flop * rows / sec / 1e12
                rec["roofline"] = {"bound": "mfma", "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS,
                                   "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA_TFLOPS, "traffic": traffic,
                                   "kernel": {"encode": extra["encode_kernel"], "opq_encode": "k_rotate_pblock5 + " + extra["encode_kernel"],
                                              "kmeans": extra["encode_kernel"] + " + k_km_{hist,scan,scatter,segsum} + codebook prep (whole iteration)",
                                              "opq_train": "k_rotate_pblock5 + 2 x " + extra["encode_kernel"] + " + k_km_* + k_reconstruct + k_atb_blocks/fold (whole step)"}[args.workload],
                                   "avg_launch_ms": kernel_ms, "algorithmic_flop_per_vector": flop,
                                   "algorithmic_bytes_per_vector": BYTES_PER_VEC,
                                   "hbm_gbs": BYTES_PER_VEC * rows / sec / 1e9,
                                   "hbm_frac": BYTES_PER_VEC * rows / sec / 1e9 / PEAK_HBM_GBS}
            if world == 1 and not args.no_cpu_baseline and args.workload in ("encode", "opq_encode"):
                rec["cpu_baseline"] = cpu_baseline(args, q, P, src, dst, pq)
            if world == 1 and not args.no_cpu_baseline and args.workload == "reconstruct":
                # BASELINE.md B-rec: the oracle's gather (primitives.rs:110-173 semantics, single thread as in
                # the reference) on the first 1 M codes, and the GPU rows checked against it byte for byte
                from oracle import pq_oracle as orc
                n_s = min(1_000_000, rows)
                c_host = src[:n_s].cpu().numpy()
                t = time.perf_counter()
                want = orc.reconstruct_batch(q, c_host)
                t_cpu = time.perf_counter() - t
                rec["cpu_baseline"] = {"value": n_s / t_cpu, "unit": "vectors/s", "cores": 1, "kind": "port",
                                       "sample": "first %d code rows of the bench batch, oracle gather on one thread (%.1f s)" % (n_s, t_cpu),
                                       "gpu_rows_identical_on_sample": bool(dst[:n_s].cpu().numpy().tobytes() == want.tobytes())}
            if world == 1 and not args.no_cpu_baseline and args.workload == "kmeans":
                rec["cpu_baseline"] = cpu_baseline_kmeans(args, q0, src, ctx)
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(args, q, P, src, dst, pq):
    """The oracle (a C port of the reference's path, CANON-F32) timed on this box's host cores on
    a bounded sample of the SAME workload, plus a parity check of the GPU codes on that sample.
    The oracle is used here only as the thing timed/checked -- never as the product."""
    from oracle import pq_oracle as orc
    cores = os.cpu_count() or 1
    n_mt = min(args.cpu_rows, src.shape[0])
    n_st = min(max(n_mt // 8, 1), 250_000)
    x = src[:n_mt].cpu().numpy()
    t = time.perf_counter()
    c_mt = orc.quantize_batch(q, x, projection=P, n_threads=cores)
    t_mt = time.perf_counter() - t
    t = time.perf_counter()
    orc.quantize_batch(q, x[:n_st], projection=P, n_threads=1)
    t_st = time.perf_counter() - t
    same = bool((dst[:n_mt].cpu().numpy() == c_mt).all())
    # PCIe-inclusive rate of the host-buffer entry point on the same sample (never `value`)
    n_h = min(n_mt, 1_000_000)
    pq.quantize_batch(x[:65536])
    t = time.perf_counter()
    c_h = pq.quantize_batch(x[:n_h])
    t_h = time.perf_counter() - t
    # the same path written as BLAS sgemm + argmin (numpy / OpenBLAS, all cores) -- a
    # stand-in for a reference linked against an optimised BLAS; not bit-comparable (its own rounding order)
    blas = None
    if P is None:
        try:
            import numpy as np
            n_b = min(n_mt, 200_000)
            xb = x[:n_b]
            cc = (q.astype(np.float32) ** 2).sum(-1)
            t = time.perf_counter()
            codes_b = np.empty((n_b, q.shape[0]), np.uint8)
            for m_ in range(q.shape[0]):
                xs = xb[:, m_ * q.shape[2]:(m_ + 1) * q.shape[2]]
                dist = (xs * xs).sum(1, keepdims=True) + cc[m_][None, :] - 2.0 * (xs @ q[m_].T)
                codes_b[:, m_] = dist.argmin(1)
            blas = {"value": n_b / (time.perf_counter() - t), "rows": n_b,
                    "agreement_with_canon": float((codes_b == c_mt[:n_b]).mean())}
        except Exception:
            blas = None
    return {"value": n_mt / t_mt, "unit": "vectors/s", "cores": cores, "kind": "port",
            "blas_formulation": blas,
            "sample": "first %d rows of the bench batch, oracle sharded over %d threads (%.1f s); "
                      "single-thread: %d rows" % (n_mt, cores, t_mt, n_st),
            "single_thread_value": n_st / t_st, "simd": "avx2+fma" if orc.lib().pqo_uses_fma_simd() else "scalar",
            "gpu_codes_identical_on_sample": same,
            "host_resident_api_value": n_h / t_h,
            "host_resident_api_identical": bool((c_h == c_mt[:n_h]).all())}


def cpu_baseline_kmeans(args, q0, src, ctx):
    """One kmeans_iteration of the oracle (assignment sharded over the host cores, update and loss
    sequential as in the reference) on a bounded sample, and the GPU result on the same sample
    checked bit for bit (centroids and losses)."""
    from oracle import pq_oracle as orc
    from reductive_amd.pq import kmeans_iterations
    cores = os.cpu_count() or 1
    n_s = min(args.cpu_rows // 2, src.shape[0])
    x = src[:n_s].cpu().numpy()
    t = time.perf_counter()
    want_q, want_loss = orc.kmeans_iterations(q0, x, n_iterations=1, n_threads=cores)
    t_cpu = time.perf_counter() - t
    got_q, got_loss = kmeans_iterations(q0, src[:n_s], n_iterations=1, want_loss=True, ctx=ctx)
    return {"value": n_s / t_cpu, "unit": "vectors/s", "cores": cores, "kind": "port",
            "sample": "one iteration over the first %d rows of the bench batch; oracle assignment on %d "
                      "threads, update + loss sequential (%.1f s)" % (n_s, cores, t_cpu),
            "gpu_centroids_identical_on_sample": bool(got_q.tobytes() == want_q.tobytes()),
            "gpu_loss_identical_on_sample": bool(got_loss.tobytes() == want_loss.tobytes())}


if __name__ == "__main__":
    main()
