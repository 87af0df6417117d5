#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native RNS polynomial arithmetic path.

Metric (BASELINE.json): CKKS N=2^16 mul+relin+rescale ops/s; batched NTT GB/s vs the HBM roofline.

  * `value`    : whole-job ops/s of multiply_inplace + relinearize_inplace + rescale_to_next on CKKS
                 ciphertexts at N=2^16 with 24 ciphertext limbs + 1 special prime (BASELINE config 4's
                 parameters), one "step" = one pass over a batch of independent ciphertext pairs that is
                 already resident in HBM.  Ranks hold independent batches (weak scaling, no data-path
                 collective; RCCL only broadcasts the relinearization key once before timing).
  * `roofline` : the dominant kernel of that path, the batched forward NTT, measured on BASELINE
                 config 2 (N=2^16, 32 60-bit primes, 1024 polys = 17.18 GB) with HIP events on the
                 launch stream; algorithmic bytes = 16 B/coefficient/transform.
  * `cpu_baseline`: the repo's CPU oracle (a restatement of SEAL's CPU path, kind "port") timed on
                 the host cores on a bounded sample of the same op, rank 0 at N=1 only.
  * parity     : rank 0 compares EVERY ciphertext of its batch with the oracle, bit for bit, at any N.
  * `config5`  : (N > 1, or --config5) BASELINE config 5: 8192 independent CKKS N=2^15 mul+relin pairs,
                 sharded contiguously over the ranks (cat_amd.sharding.shard_range), strong scaling.

Usage: python bench.py [--gpus N] [--steps K] [--warmup W]
  --gpus N > 1 without a launcher (no WORLD_SIZE in the environment): this process starts N rank processes of
  itself (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_ADDR/MASTER_PORT set) BEFORE anything touches the GPU, waits for
  them and relays rank 0's JSON line.  Under torchrun (WORLD_SIZE set) it is one of the ranks.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LOG_N = 16
N = 1 << LOG_N
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
METRIC = "CKKS N=2^16 mul+relin+rescale ops/sec; batched NTT GB/s vs HBM roofline"


def get_primes_desc(factor, bits, count):
    """util::get_primes (numth.cpp:278-311) -- host-side restatement for the synthetic parameter sets."""
    def is_prime(n):
        if n < 2:
            return False
        for p in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
            if n % p == 0:
                return n == p
        d, r = n - 1, 0
        while d % 2 == 0:
            d //= 2
            r += 1
        for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
            x = pow(a, d, n)
            if x in (1, n - 1):
                continue
            for _ in range(r - 1):
                x = x * x % n
                if x == n - 1:
                    break
            else:
                return False
        return True

    out = []
    v = ((1 << bits) - 1) // factor * factor + 1
    while len(out) < count:
        if is_prime(v):
            out.append(v)
        v -= factor
    return out


def bfv_default_32768():
    """CoeffModulus::BFVDefault(32768) (util/globals.cpp:61-70): 15 55-bit primes, largest first, then one of 56 bits."""
    return get_primes_desc(2 * 32768, 55, 15) + [get_primes_desc(2 * 32768, 56, 1)[0]]


def uniform_limbs(torch, primes, prefix, n, device, gen, chunk=64):
    """[*prefix, len(primes), n] int64 tensor, limb j uniform in [0, q_j); generated `chunk` leading items at a time"""
    out = torch.empty(tuple(prefix) + (len(primes), n), dtype=torch.int64, device=device)
    lead = prefix[0] if prefix else 1
    for b0 in range(0, lead, chunk):
        sl = out[b0:b0 + chunk] if prefix else out
        for j, q in enumerate(primes):
            sl[..., j, :] = torch.randint(0, q, tuple(sl.shape[:-2]) + (n,), dtype=torch.int64, device=device,
                                          generator=gen)
    return out


# ---- launcher: N rank processes of this file, started before any GPU call ------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n_ranks, argv):
    """Starts `n_ranks` fresh processes of this script (never an exec of a process that has touched the GPU: this
    parent imports neither torch nor cat_amd), relays rank 0's stdout, fails if any rank fails or if the emitted
    line does not say n_gpus == n_ranks."""
    port = free_port()
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True if r == 0 else None))
    out_lines = []

    def pump():
        for line in procs[0].stdout:
            out_lines.append(line)

    th = threading.Thread(target=pump, daemon=True)
    th.start()
    rc = 0
    live = set(range(n_ranks))
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code
                for o in live:  # exactly the processes started above, by PID
                    procs[o].terminate()
        time.sleep(0.05)
    th.join(timeout=10)
    line = None
    for ln in out_lines:
        if ln.lstrip().startswith("{"):
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    if rc != 0:
        if line:
            print(line)
        sys.stderr.write(f"bench.py launcher: a rank exited with code {rc}\n")
        return rc
    if line is None:
        sys.stderr.write("bench.py launcher: rank 0 printed no JSON line\n")
        return 1
    if json.loads(line).get("n_gpus") != n_ranks:
        sys.stderr.write(f"bench.py launcher: the emitted line does not say n_gpus == {n_ranks}\n")
        print(line)
        return 1
    print(line)
    return 0


def launcher_selftest():
    """Rank body of `--launcher-selftest`: the rendezvous, barrier and max-over-ranks of the real bench over gloo on
    the CPU, no GPU and no measurement (tests/test_bench_launcher.py)."""
    import torch
    import torch.distributed as dist

    from cat_amd import sharding

    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    if os.environ.get("CAT_BENCH_SELFTEST_FAIL_RANK") == str(rank):
        sys.exit(3)  # the test of the launcher's failure path: the other ranks are left waiting in the rendezvous
    dist.init_process_group("gloo")
    dist.barrier()
    slowest = sharding.max_over_ranks(float(rank + 1))
    begin, end = sharding.shard_range(8192, world, rank)
    sizes = torch.tensor([end - begin], dtype=torch.int64)
    dist.all_reduce(sizes)
    if rank == 0:
        print(json.dumps({"metric": METRIC, "selftest": True, "n_gpus": world, "max_over_ranks": slowest,
                          "config5_pairs": int(sizes.item())}))
    dist.barrier()
    dist.destroy_process_group()


def cpu_description():
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = os.cpu_count() or 1
    return model, os.cpu_count() or 1, affinity


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs (default: WORLD_SIZE under a launcher, else 1)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--settle", type=int, default=24,
                    help="untimed steps before the W warm-up steps: the card's clocks need about 40 ms of load to settle "
                         "(warm-up 5 / 10 / 20 / 40 steps: 3149 / 3189 / 3198 / 3198 ops/s on one box)")
    ap.add_argument("--batch", type=int, default=8, help="ciphertext pairs per step per GPU")
    ap.add_argument("--limbs", type=int, default=24, help="ciphertext limbs L (key level has L+1 primes)")
    ap.add_argument("--ntt-polys", type=int, default=1024)
    ap.add_argument("--ntt-limbs", type=int, default=32)
    ap.add_argument("--ntt-reps", type=int, default=5)
    ap.add_argument("--skip-ntt", action="store_true")
    ap.add_argument("--skip-cpu", action="store_true", help="skip the CPU timing (the parity check stays)")
    ap.add_argument("--skip-parity", action="store_true")
    ap.add_argument("--skip-latency", action="store_true", help="skip the single-ciphertext latency table (N=32768)")
    ap.add_argument("--cpu-reps", type=int, default=10, help="CPU baseline ops per host thread")
    ap.add_argument("--cpu-threads", type=int, default=0, help="CPU baseline threads (0 = every core this process may use)")
    ap.add_argument("--config5", action="store_true", help="run the BASELINE config 5 leg also at N=1 (193 GB resident)")
    ap.add_argument("--config5-pairs", type=int, default=8192)
    ap.add_argument("--config5-slice", type=int, default=64)
    ap.add_argument("--launcher-selftest", action="store_true", help=argparse.SUPPRESS)
    # rehearsal of the N > 1 code path on a one-GPU box: every rank on cuda:0, collectives over gloo (RCCL refuses two
    # ranks on one device); the numbers of such a run mean nothing and the line says so
    ap.add_argument("--rehearse-on-one-gpu", action="store_true", help=argparse.SUPPRESS)
    # one process, one-rank communicator: drives the guarded product path below on a one-GPU box (tests/test_gpu_comm.py)
    ap.add_argument("--rccl-selftest", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--rccl-timeout", type=float, default=120.0,
                    help="seconds the product's RCCL key broadcast may take (N > 1) before torch.distributed takes over")
    args = ap.parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus is None:  # `torchrun --nproc-per-node N bench.py` without --gpus: the launcher's word counts
        args.gpus = int(env_world) if env_world else 1
    if args.gpus < 1:
        ap.error("--gpus")

    if env_world is None and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(env_world or "1")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(json.dumps({"error": f"--gpus {args.gpus} but the launcher started {world} ranks"}))
        sys.exit(2)
    if args.launcher_selftest:
        launcher_selftest()
        return

    # The single-ciphertext latency table is a program of its own (tools/latency_facade, C++ over the C ABI): run it as a
    # child process now, before this process has touched the GPU -- the card is idle for it and no GPU-initialised
    # process ever starts another program.
    latency = None
    if rank == 0 and world == 1 and not args.skip_latency:
        latency = latency_table()

    import numpy as np
    import torch
    import torch.distributed as dist

    import cat_amd
    from cat_amd import sharding
    from cat_amd.api import to_host

    if args.rehearse_on_one_gpu:
        import faulthandler

        local_rank = 0
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        # a rehearsal that hangs says where; the limit is the one tests/test_zy_bench_two_ranks.py gives the whole run
        faulthandler.dump_traceback_later(280, exit=True)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    torch.cuda.set_device(local_rank)
    device = torch.device(f"cuda:{local_rank}")
    gen = torch.Generator(device=device)
    gen.manual_seed(1234 + rank)

    L = args.limbs
    k = L + 1
    # CoeffModulus::Create hands out equal-size primes smallest first (modulus.cpp:177-183)
    primes = list(reversed(get_primes_desc(2 * N, 60, k)))
    ctx = cat_amd.Context(LOG_N, primes, cat_amd.SCHEME_CKKS, device=local_rank)

    # ---- synthetic inputs, resident in HBM before the timed region -------------------------------------
    B = args.batch
    x = uniform_limbs(torch, primes[:L], (B, 2), N, device, gen)
    y = uniform_limbs(torch, primes[:L], (B, 2), N, device, gen)
    key = torch.empty((k - 1, 2, k, N), dtype=torch.int64, device=device)
    if rank == 0:
        kgen = torch.Generator(device=device)
        kgen.manual_seed(99)
        key.copy_(uniform_limbs(torch, primes, (k - 1, 2), N, device, kgen))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    # The only collective, outside the timed region: the product's own cat_key_broadcast (C ABI) on an RCCL communicator
    # made here (one rank per GPU over xGMI).  The gloo rehearsal on one GPU keeps torch.distributed's broadcast: RCCL
    # refuses two ranks on one device.
    # Its first run on more than one GPU is the driver's scaling run, so it is guarded: communicator set-up and the
    # broadcast run in a worker thread with a time limit, the ranks then agree (all_reduce MIN over the process group)
    # whether every one of them got through, and if not the key goes out through torch.distributed instead -- the
    # bench line says which way it went and why.
    key_bcast_s = 0.0
    comm = None  # the communicator, once it has carried a broadcast on every rank
    if (world > 1 or args.rccl_selftest) and not args.rehearse_on_one_gpu:
        box = {}

        def product_path():
            try:
                torch.cuda.set_device(local_rank)  # the current device is per thread
                comm = sharding.RcclComm(world, rank)
                box["comm"] = comm
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                ctx.key_broadcast(key, comm.handle, root=0)
                torch.cuda.synchronize()
                box["s"] = time.perf_counter() - t0
            except Exception as e:  # noqa: BLE001 -- reported in the bench line, the fallback takes over
                box["error"] = f"{type(e).__name__}: {e}"

        worker = threading.Thread(target=product_path, daemon=True)
        worker.start()
        worker.join(timeout=args.rccl_timeout)
        mine_ok = (not worker.is_alive()) and "s" in box
        flag = torch.tensor([1 if mine_ok else 0], dtype=torch.int32, device=device)
        if world > 1:
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            key_bcast_s = box["s"]
            comm = box["comm"]
            key_bcast_via = "cat_key_broadcast (RCCL communicator of the bench's own)"
        else:
            why = box.get("error") or ("timed out" if worker.is_alive() else "another rank failed")
            t0 = time.perf_counter()
            sharding.broadcast_key(key, src=0)
            torch.cuda.synchronize()
            key_bcast_s = time.perf_counter() - t0
            key_bcast_via = f"torch.distributed broadcast (cat_key_broadcast did not complete on every rank; rank {rank}: {why})"
    else:
        t0 = time.perf_counter()
        sharding.broadcast_key(key, src=0)
        torch.cuda.synchronize()
        key_bcast_s = time.perf_counter() - t0 if world > 1 else 0.0
        key_bcast_via = "torch.distributed broadcast (gloo rehearsal)" if world > 1 else "single process: no collective"
    key_bcast_s = sharding.max_over_ranks(key_bcast_s, device) if world > 1 else key_bcast_s
    prod = torch.empty((B, 3, L, N), dtype=torch.int64, device=device)
    out = torch.empty((B, 2, L - 1, N), dtype=torch.int64, device=device)

    two_calls = os.environ.get("CAT_BENCH_TWO_CALLS") is not None  # A/B: relinearize and rescale as separate calls

    def step():
        ctx.ckks_multiply(x, y, out=prod)          # Evaluator::multiply_inplace   (evaluator.cpp:569)
        if two_calls:
            ctx.relinearize(prod, key)             # Evaluator::relinearize_inplace (evaluator.cpp:1144)
            # the reference's resize to size 2 (evaluator.cpp:1191) is a no-op on the buffer: rescale reads the
            # first two polynomials of every size-3 slot in place
            ctx.rescale_strided(prod, 2, out)      # Evaluator::rescale_to_next    (evaluator.cpp:1481)
        else:
            # the same two reference calls through the fused entry point: identical words (checked against the
            # oracle's relinearize + rescale below), one forward transform per limb instead of two
            ctx.relinearize_rescale(prod, key, out=out)

    for _ in range(max(0, args.settle)):  # untimed, like the warm-up: first launches, workspace, clocks
        step()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed = sharding.max_over_ranks(elapsed, device)
    ops_per_s = world * B * args.steps / elapsed
    # algorithmic bytes per op: tensor 56LN + key switch 8N[L+2L(L+1)+4L] + rescale 2(2L-1)N8
    bytes_per_op = 56 * L * N + 8 * N * (L + 2 * L * (L + 1) + 4 * L) + 2 * (2 * L - 1) * N * 8

    result = {
        **({"rehearsal": "all ranks on one GPU over gloo: code-path check only, not a measurement"}
           if args.rehearse_on_one_gpu else {}),
        "metric": METRIC,
        "value": ops_per_s,
        "unit": "ops/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {
            "workload": f"CKKS N=65536, L={L} ciphertext limbs + 1 special 60-bit prime: multiply + relinearize + "
                        f"rescale_to_next on {B} independent ciphertext pairs per GPU per step (BASELINE config 4 "
                        "parameters, metric op)",
            "batch_per_gpu": B,
            "settle_steps": max(0, args.settle),  # untimed steps run before the W warm-up steps (clock settling)
            "calls": "cat_ckks_multiply + cat_relinearize + cat_rescale_strided" if two_calls
                     else "cat_ckks_multiply + cat_relinearize_rescale (the words of relinearize then rescale_to_next)",
            "parallelism": f"batch-sharded x{world}, no data-path collective",
            "algorithmic_bytes_per_op": bytes_per_op,
            "op_hbm_floor_frac": (ops_per_s / world) * bytes_per_op / (HBM_PEAK_GBS * 1e9),
            "key_broadcast_s": key_bcast_s,
            "key_broadcast_via": key_bcast_via,
        },
    }

    # ---- bit-exact check of rank 0's whole batch (any N) + CPU baseline (N=1 only) ---------------------------
    if rank == 0 and not args.skip_parity:
        from oracle import seal_oracle as so

        so.build()
        octx = so.Context(LOG_N, primes, so.SCHEME_CKKS)
        hx, hy, hkey, hout = to_host(x), to_host(y), to_host(key), to_host(out)
        model, logical, usable = cpu_description()
        env_cap = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)  # the GPU box states its CPU share this way
        threads = args.cpu_threads or (min(usable, env_cap) if env_cap else usable)
        threads = max(1, threads)

        def one_op(i):
            return octx.rescale(octx.relinearize(octx.ckks_multiply(hx[i], hy[i]), hkey))

        # every ciphertext of the batch, B oracle ops on up to `threads` threads (ctypes releases the GIL)
        match = [False] * B

        def check(lo, hi):
            for i in range(lo, hi):
                match[i] = bool((hout[i] == one_op(i)).all())

        t1 = time.perf_counter()
        single = one_op(0)  # one op alone on one thread: SEAL itself runs an op on one thread
        single_alone_s = time.perf_counter() - t1
        match[0] = bool((hout[0] == single).all())
        del single
        nthr = min(threads, max(1, B - 1))
        bounds = [1 + (B - 1) * i // nthr for i in range(nthr + 1)]
        ths = [threading.Thread(target=check, args=(bounds[i], bounds[i + 1])) for i in range(nthr)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        ok = all(match)
        result["parity_checked"] = ok
        result["parity"] = f"{sum(match)} of {B} ciphertexts of rank 0's batch equal the CPU oracle bit for bit"
        if not ok:
            print(json.dumps({"error": "GPU result differs from the CPU oracle", **result}))
            sys.exit(1)

        if world == 1 and not args.skip_cpu:
            reps = args.cpu_reps  # ops per thread: ~1.5 s each on one core -> 10-20 s of wall time

            def cpu_op(i):
                for _ in range(reps):
                    one_op(i % B)

            t1 = time.perf_counter()
            ths = [threading.Thread(target=cpu_op, args=(i,)) for i in range(threads)]
            for th in ths:
                th.start()
            for th in ths:
                th.join()
            cpu_s = time.perf_counter() - t1
            # BASELINE config 1 (the reference's own CPU-runnable case): N=8192, {60,40,40,60}, one thread
            p1 = so.coeff_modulus_create(8192, [60, 40, 40, 60])
            o1 = so.Context(13, p1, so.SCHEME_CKKS)
            sk1 = o1.secret_key(1)
            rk1 = o1.relin_key(sk1, 2)
            a1, b1 = o1.uniform(3, 2, 3), o1.uniform(3, 2, 4)
            o1.rescale(o1.relinearize(o1.ckks_multiply(a1, b1), rk1))
            t1 = time.perf_counter()
            c1_reps = 20
            for _ in range(c1_reps):
                o1.rescale(o1.relinearize(o1.ckks_multiply(a1, b1), rk1))
            c1_s = (time.perf_counter() - t1) / c1_reps
            result["cpu_baseline"] = {
                "value": threads * reps / cpu_s,
                "unit": "ops/s",
                "cores": threads,
                "kind": "port",
                "cpu_model": model,
                "host_logical_cpus": logical,
                "usable_cpus": usable,
                "single_thread_ops_per_s": 1.0 / single_alone_s,
                "config1_ops_per_s": 1.0 / c1_s,
                "sample": f"{threads * reps} x (multiply+relinearize+rescale) of the batch's ciphertext pairs, {reps} ops on "
                          f"each of {threads} threads (ctypes releases the GIL; {threads} = the CPUs this process may use: "
                          f"affinity {usable} of {logical} logical CPUs"
                          + (f", OMP_NUM_THREADS={env_cap} is the box's CPU share" if env_cap else "") +
                          f"), {cpu_s:.1f} s wall; config1_ops_per_s = BASELINE config 1 (N=8192, primes 60/40/40/60, "
                          f"multiply+relinearize+rescale) on one thread, {c1_reps} ops",
            }
            del o1
        del octx

    # ---- BASELINE config 5: 8192 independent CKKS N=2^15 multiply + relinearize, batch-sharded -----------------
    if world > 1 or args.config5:
        del x, y, prod, out, key
        ctx.close()
        torch.cuda.empty_cache()
        result["config5"] = config5_leg(args, torch, dist, cat_amd, sharding, to_host, world, rank, local_rank, device, gen,
                                        comm)
        # rank 0 holds the verdict: every rank learns it and exits non-zero with it
        bad = 1.0 if result["config5"].get("parity_checked") is False else 0.0
        if sharding.max_over_ranks(bad, device) > 0:
            if rank == 0:
                print(json.dumps({"error": "config 5: GPU result differs from the CPU oracle", **result}))
            if world > 1:
                dist.destroy_process_group()
            sys.exit(1)
        x = y = prod = out = None

    # ---- roofline leg: batched forward/inverse NTT, BASELINE config 2 --------------------------------------
    if rank == 0 and not args.skip_ntt:
        x = y = prod = out = key = None
        ctx.close()
        torch.cuda.empty_cache()
        nprimes = get_primes_desc(2 * N, 60, args.ntt_limbs)
        nctx = cat_amd.Context(LOG_N, nprimes, cat_amd.SCHEME_NONE, device=local_rank)
        P = args.ntt_polys
        data = uniform_limbs(torch, nprimes, (P,), N, device, gen)
        ref = data.clone()
        coeffs = P * args.ntt_limbs * N
        alg_bytes = 16 * coeffs
        fwd_ms, inv_ms = [], []
        for rep in range(args.ntt_reps + 1):
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record()
            nctx.ntt_forward(data)
            e1.record()
            nctx.ntt_inverse(data)
            e2.record()
            torch.cuda.synchronize()
            if rep:  # first repetition is the warm-up
                fwd_ms.append(e0.elapsed_time(e1))
                inv_ms.append(e1.elapsed_time(e2))
        roundtrip = bool(torch.equal(data, ref))
        # what a plain device-to-device copy of the same bytes reaches on this box (read + write counted)
        ce0, ce1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ref.copy_(data)
        ce0.record()
        for _ in range(3):
            ref.copy_(data)
        ce1.record()
        torch.cuda.synchronize()
        copy_gbps = 3 * 2 * data.numel() * 8 / (ce0.elapsed_time(ce1) * 1e-3) / 1e9
        f_ms, i_ms = float(np.mean(fwd_ms)), float(np.mean(inv_ms))
        achieved = alg_bytes / (f_ms * 1e-3) / 1e9
        traffic, traffic_src, kernels = None, None, None
        pmc = os.path.join(ROOT, "profiles", "ntt_pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))
                traffic = rec.get("hbm_bytes_per_forward_transform")
                traffic_src = rec.get("collected")
                kernels = rec.get("forward_kernels")
            except Exception:
                traffic = None
        result["roofline"] = {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": traffic_src,  # PMC passes are separate runs (profiles/ntt_pmc_traffic.json), not this one
            "kernel": "batched forward NTT" + (f" = {kernels}" if kernels else "") +
                      " (`achieved` = the transform's algorithmic bytes over its event-timed duration)",
            "workload": f"N=65536, {args.ntt_limbs} x 60-bit primes, {P} polys ({coeffs * 8 / 1e9:.2f} GB), in place",
            "algorithmic_bytes_per_transform": alg_bytes,
            "forward_ms": f_ms,
            "inverse_ms": i_ms,
            "inverse_GBps": alg_bytes / (i_ms * 1e-3) / 1e9,
            "roundtrip_exact": roundtrip,
            "copy_GBps": copy_gbps,  # measured streaming copy (bytes read + written)
            # the transform makes two passes over its data: two plain device copies of it against the transform's time
            "two_copies_over_forward": (2 * alg_bytes / (copy_gbps * 1e9)) / (f_ms * 1e-3),
        }
        if not roundtrip:
            print(json.dumps({"error": "NTT round trip mismatch", **result}))
            sys.exit(1)

    # ---- single-ciphertext latencies at the reference's published configuration (CKKS N=32768, 16 primes) ---------
    if latency is not None:
        result["config"]["latency_us"] = latency

    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.barrier()  # rank 0 may still be in its roofline leg
    if comm is not None:
        comm.close()
    if world > 1:
        dist.destroy_process_group()


# image.png of the reference (microseconds per call, CKKS N=32768, 16 primes, one ciphertext): the
# only numbers it publishes.  Other hardware -- context for the table below, not a same-node comparison.
REFERENCE_PUBLISHED_US = {
    "SEAL CPU": {"add": 3954, "multiply": 8187, "multiply_plain": 2773, "square": 8269, "relinearize": 175520,
                 "rescale": 20132, "rotate_vector_one_step": 175741, "rotate_vector_random": 876642,
                 "complex_conjugate": 174508},
    "CAT RTX 3090": {"add": 8, "multiply": 34, "multiply_plain": 6, "square": 26, "relinearize": 4216, "rescale": 696,
                     "rotate_vector_one_step": 4264, "rotate_vector_random": 20084, "complex_conjugate": 4109},
    "CAT RTX 4090": {"multiply": 15, "multiply_plain": 4, "relinearize": 2472, "rescale": 441,
                     "rotate_vector_one_step": 2218},
}


def latency_table():
    """Runs tools/latency_facade (seal::Evaluator of the C++ facade, one ciphertext per call, the loop of the
    reference's native/examples/6_performance.cpp:348-603) as a child process and returns its medians."""
    exe = os.path.join(ROOT, "tools", "latency_facade")
    if not os.path.exists(exe):
        return {"error": "tools/latency_facade is not built (make latency_facade)"}
    try:
        r = subprocess.run([exe, "10"], capture_output=True, text=True, timeout=240)
        rec = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:  # the table is an extra: its failure must not take the bench line with it
        return {"error": f"tools/latency_facade failed: {e}"}
    if "error" in rec:
        return rec
    return {
        "workload": rec["workload"],
        "through": "seal::Evaluator (cat_amd/host/seal_hip.hpp) over the C ABI, default stream",
        "timing": rec["timing"] + f"; median of {rec['count']} after one warm-up",
        "evaluator_median_us": rec["evaluator_median_us"],
        "c_abi_batch1_median_us": rec["c_abi_batch1_median_us"],
        "workspace_grew": rec["workspace_grew"],
        "reference_published_us_other_hardware": REFERENCE_PUBLISHED_US,
    }


def config5_leg(args, torch, dist, cat_amd, sharding, to_host, world, rank, local_rank, device, gen, comm=None):
    """8192 (ct_a, ct_b) pairs, CKKS N=32768, BFVDefault(32768) = 15 ciphertext primes + special; rank r owns the
    contiguous slice shard_range(8192, world, r), resident in HBM, and runs multiply + relinearize over it in slices
    of `--config5-slice` pairs; one shared relinearization key, broadcast once.  Strong scaling: the job is fixed."""
    import time as _t

    log_n, n, L = 15, 32768, 15
    primes = bfv_default_32768()
    ctx = cat_amd.Context(log_n, primes, cat_amd.SCHEME_CKKS, device=local_rank)
    total = args.config5_pairs
    begin, end = sharding.shard_range(total, world, rank)
    mine, S = end - begin, args.config5_slice
    x = uniform_limbs(torch, primes[:L], (mine, 2), n, device, gen)
    y = uniform_limbs(torch, primes[:L], (mine, 2), n, device, gen)
    res = torch.empty((mine, 3, L, n), dtype=torch.int64, device=device)  # relinearize leaves the result in polys 0, 1
    key = torch.empty((len(primes) - 1, 2, len(primes), n), dtype=torch.int64, device=device)
    if rank == 0:
        kgen = torch.Generator(device=device)
        kgen.manual_seed(55)
        key.copy_(uniform_limbs(torch, primes, (len(primes) - 1, 2), n, device, kgen))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = _t.perf_counter()
    if comm is not None:
        ctx.key_broadcast(key, comm.handle, root=0)  # the C ABI's collective, as in the metric leg
    else:
        sharding.broadcast_key(key, src=0)
    torch.cuda.synchronize()
    bcast_s = _t.perf_counter() - t0 if world > 1 else 0.0

    def run():
        for b0 in range(0, mine, S):
            ctx.ckks_multiply(x[b0:b0 + S], y[b0:b0 + S], out=res[b0:b0 + S])
            ctx.relinearize(res[b0:b0 + S], key)

    # warm-up on the first slice only (sizes the workspace); the timed pass covers the whole shard once
    if mine:
        ctx.ckks_multiply(x[:S], y[:S], out=res[:S])
        ctx.relinearize(res[:S], key)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = _t.perf_counter()
    run()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = sharding.max_over_ranks(_t.perf_counter() - t0, device)
    leg = {
        "workload": f"BASELINE config 5: {total} independent CKKS N=32768 (15 + special prime, BFVDefault) multiply + "
                    f"relinearize pairs, rank r owns shard_range({total}, {world}, r), slices of {S}",
        "pairs": total,
        "pairs_rank0": mine,
        "ops_per_s": total / elapsed,
        "seconds": elapsed,
        "scaling": "strong",
        "n_gpus": world,
        "key_broadcast_s": bcast_s,
        "key_bytes": key.numel() * 8,
    }
    if rank == 0 and mine and not args.skip_parity:
        from oracle import seal_oracle as so

        octx = so.Context(log_n, primes, so.SCHEME_CKKS)
        hkey = to_host(key)
        ok = True
        for i in sorted({0, mine // 2, mine - 1}):
            exp = octx.relinearize(octx.ckks_multiply(to_host(x[i]), to_host(y[i])), hkey)
            ok = ok and bool((to_host(res[i, :2]) == exp).all())
        leg["parity_checked"] = ok
    ctx.close()
    return leg


if __name__ == "__main__":
    main()
