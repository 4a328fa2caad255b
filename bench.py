#!/usr/bin/env python3
"""Headline benchmark: 512x512 4-step LCM images/sec on N MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" is one pass of the hot path over one batch: a full 4-step LCM sample (4 x UNet + scheduler steps +
VAE decode to RGB8) replayed from the captured hipGraph, inputs (prompt embeddings, per-request noise)
already resident in HBM.  Workload at N=1 = BASELINE.json configs[1]: SD1.5 512x512, 4 steps, batch 1, fp16,
hipGraph.  Independent requests shard across ranks with no data-path collective (weak scaling, fixed
per-GPU batch); the only exchange is one RCCL broadcast of the prompt embeddings before the timed region.
Weights are seeded synthetic SD1.5-architecture weights (no checkpoint ships with the reference).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_F16_PEAK_TFLOPS = 2500.0   # MI355X dense fp16/bf16 MFMA peak (MI355X_MICROARCH.md, chip-level parameters)
HBM_PEAK_GBS = 8000.0


def host_threads():
    """Host cores this process may use (a 1-GPU box shares 16 of the host's cores per GPU)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    return max(1, min(n, int(os.environ.get("LCM_CPU_THREADS", "16"))))


def cpu_baseline(threads):
    """Oracle (CPU restatement, kind 'port') on a bounded sample of the same workload: ONE whole 512x512 4-step image
    through LCMPipelineOracle (latents from the seed -> 4 x [UNet -> LCMScheduler.step] -> VAE decode -> u8), after one
    warm-up UNet forward (thread pool, allocator)."""
    import numpy as np
    import torch
    from sdlcm_amd import weights
    from oracle.pipeline import LCMPipelineOracle
    from oracle import glue
    torch.set_num_threads(threads)
    ora = LCMPipelineOracle(weights.synthetic_unet(), weights.synthetic_vae())
    g = torch.Generator().manual_seed(0)
    pe = torch.randn(1, 77, 768, generator=g)
    cond = torch.from_numpy(glue.guidance_scale_embedding(np.zeros(1, np.float32), 256))
    with torch.inference_mode():
        ora.unet.forward(torch.randn(1, 4, 64, 64, generator=g), 999, pe, cond)          # warm-up
    t0 = time.time()
    out = ora(pe, 512, 512, 4, 1.0, 1000)
    t_img = time.time() - t0
    assert out["image_u8"].size == 512 * 512 * 3
    return {"value": round(1.0 / t_img, 5), "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"1 whole 512x512 4-step image (seed 1000) through oracle.pipeline.LCMPipelineOracle after one warm-up "
                      f"UNet forward: {t_img:.2f} s, torch-CPU fp32, {threads} threads"}


def roofline_leg(pipe, P, guidance, ms_per_step):
    """Live HIP-event timing of the dominant MFMA kernel.

    Pass 1 (attribution): one eager pass of the same workload with the library bracketing every MFMA launch by HIP
    events on its launch stream (kernel instantiation names as rocprofv3 reports them).  An event pair costs a few us
    of queue serialisation per launch, so these per-launch times only rank the kernels (and are reported as
    `bracketed_*`).  Pass 2 (the reported number): every launch of the dominant instantiation, with its own layer's
    operands and weights, is replayed back to back behind graph replays (busy stream, hot clocks) inside ONE event
    pair; avg = elapsed / launches.  achieved = sum of algorithmic FLOP of those launches / elapsed."""
    import torch
    from sdlcm_amd import ops
    with torch.cuda.stream(pipe.stream), ops.profiling() as recs, ops.recording() as closures:
        for _ in range(max(2, int((150.0 + 120.0 * P.B) / max(ms_per_step, 1e-3)) + 1)):
            P.graph.launch()
        ops.profile_begin()
        pipe._enqueue(P, guidance)
        pipe.stream.synchronize()
        times = ops.profile_end()
    assert len(recs) == len(times) == len(closures), (len(recs), len(times), len(closures))
    agg = {}
    for r, (name, ms), c in zip(recs, times, closures):
        a = agg.setdefault(name, dict(n=0, flops=0.0, bytes=0.0, ms=0.0, kind=r["kind"], fns=[]))
        a["n"] += 1; a["flops"] += r["flops"]; a["bytes"] += r.get("bytes", 0.0); a["ms"] += ms; a["fns"].append(c[2])
    name, dom = max(agg.items(), key=lambda kv: kv[1]["ms"])
    with torch.cuda.stream(pipe.stream):
        for fn in dom["fns"]:
            fn()
        pipe.stream.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(max(2, int(40.0 * dom["n"] / 1000.0 / max(ms_per_step, 1e-3)) + 2)):
            P.graph.launch()
        e0.record()
        for fn in dom["fns"]:
            fn()
        e1.record()
        e1.synchronize()
        iso_ms = e0.elapsed_time(e1)
    ach = dom["flops"] / (iso_ms * 1e-3) / 1e12
    table = {k: {"launches": v["n"], "bracketed_ms": round(v["ms"], 3), "bracketed_avg_us": round(v["ms"] * 1e3 / v["n"], 2),
                 "gflop": round(v["flops"] / 1e9, 1)} for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])}
    tot_f = sum(v["flops"] for v in agg.values())
    traffic, traffic_src = _pmc_traffic(name.replace(" +splitk", ""), P.B)     # rocprof names carry no split suffix
    out = {"bound": "mfma", "kernel": f"{name} ({dom['kind']})",
            "achieved": round(ach, 1), "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_F16_PEAK_TFLOPS, 4),
            "launches": dom["n"], "avg_launch_us": round(iso_ms * 1e3 / dom["n"], 2),
            "flop_per_launch_avg": round(dom["flops"] / dom["n"]), "traffic": traffic,
            "method": "dominant instantiation replayed back-to-back on its real per-layer operands inside one HIP event pair",
            "mfma_flop_per_pass": round(tot_f), "by_kernel": table}
    out["algorithmic_bytes_per_launch_avg"] = round(dom["bytes"] / dom["n"])     # operands + result, each counted once
    out.update(_pmc_counters(name.replace(" +splitk", "")))
    # `bound` names the roof `achieved` / `peak` are quoted against (a contraction: MFMA).  Which resource actually limits the
    # launch is a separate statement, made from the counters when they are fresh: neither roof within 25 % = latency
    hb = out.get("hbm_gbs")
    fr_m, fr_h = out["frac"], (hb / HBM_PEAK_GBS if hb else None)
    out["limiting"] = ("unknown (no fresh counters)" if fr_h is None else
                       "mfma" if fr_m >= 0.25 and fr_m >= fr_h else "hbm" if fr_h >= 0.25 else
                       f"latency (mfma {fr_m:.0%} and hbm {fr_h:.0%} of their roofs: a chain of short dependent launches)")
    out["traffic_unit"] = "bytes per launch (fabric reads + writes)"
    out["traffic_source"] = traffic_src
    out["kernel_launches_per_pass"] = len(times)
    return out


TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r04_traffic.json")
PMC_SUMMARY_FILE = os.path.join(ROOT, "profiles", "r04_pmc_summary.json")


def _pmc_counters(kernel_name):
    """MFMA-busy fraction and HBM GB/s of the dominant kernel from the committed rocprofv3 --pmc passes over an eager
    single-kernel target of the same shape class (profiles/r04_pmc_summary.json <- tools/pmc_round4.sh + tools/pmc_summary.py;
    counters cannot be read from inside this process).  The entry must match the instantiation UNIQUELY and must have been taken
    on the kernel sources of this tree (its recorded sha256, tools/check_profiles_fresh.py): otherwise the fields are null and the
    reason is given -- a counter of another kernel, or of an older build of this one, is never reported."""
    import json
    none = {"mfma_busy_frac": None, "hbm_gbs": None}
    try:
        with open(PMC_SUMMARY_FILE) as f:
            tab = json.load(f)
    except Exception as e:
        return dict(none, pmc_source=f"{os.path.relpath(PMC_SUMMARY_FILE, ROOT)} unreadable ({e!r})")
    key = kernel_name.strip()
    hits = [k for k in tab if k.strip() == key] or [k for k in tab if k.rstrip(">").strip().startswith(key.rstrip(">").strip() + ",") or k.strip().startswith(key)]
    if len(hits) != 1:
        return dict(none, pmc_source=f"kernel '{kernel_name}' matches {len(hits)} entries of {os.path.relpath(PMC_SUMMARY_FILE, ROOT)} "
                                     f"(has {sorted(tab)}): run tools/pmc_round4.sh")
    try:
        from tools.check_profiles_fresh import stale_entries
        stale = stale_entries(PMC_SUMMARY_FILE).get(hits[0])
    except Exception as e:
        stale = f"freshness check failed ({e!r})"
    if stale:
        return dict(none, pmc_source=f"{os.path.relpath(PMC_SUMMARY_FILE, ROOT)}['{hits[0]}'] is stale: {stale}")
    v = tab[hits[0]]
    clk = v.get("effective_clock_ghz")
    if clk is not None and clk > 2.5 and v.get("mfma_busy_frac_at_2p1ghz") is not None:
        # launches of a few microseconds: GRBM_GUI_ACTIVE also counts the dispatch around the kernel (an "effective clock"
        # above the chip's 2.4 GHz), so the busy cycles are taken over the kernel's own duration at 2.1 GHz instead
        return {"mfma_busy_frac": v["mfma_busy_frac_at_2p1ghz"], "hbm_gbs": v["hbm_gbs"], "effective_clock_ghz": None,
                "pmc_source": f"{v['source']} ({v['what']}): SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over the kernel's duration x 2.1 GHz "
                              f"(GRBM_GUI_ACTIVE over-counts launches this short); (2 x FETCH_SIZE + WRITE_SIZE) / kernel time"}
    return {"mfma_busy_frac": v["mfma_busy_frac"], "hbm_gbs": v["hbm_gbs"], "effective_clock_ghz": clk,
            "pmc_source": f"{v['source']} ({v['what']}): SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs; "
                          f"(2 x FETCH_SIZE + WRITE_SIZE) / kernel time"}


def _pmc_traffic(kernel_name, batch):
    """HBM-side bytes per launch of `kernel_name`, from the committed PMC passes (profiles/r04_traffic.json: rocprofv3 --pmc
    FETCH_SIZE and WRITE_SIZE in separate runs over one eager pass of this workload, tools/pmc_pass.sh; counters cannot be
    read from inside bench.py).  FETCH_SIZE x2 is the gfx950 correction of MI355X_MICROARCH.md's HBM section.  When the
    dominant kernel / batch is not in the file the value is None AND the reason is reported (`traffic_source`) and printed
    to stderr -- never a silent null, never a guess."""
    import json
    try:
        with open(TRAFFIC_FILE) as f:
            doc = json.load(f)
        tab = doc["batch"].get(str(batch), {})
        from tools.check_profiles_fresh import sources_sha256
        if not doc.get("csrc_sha256") or sources_sha256(doc.get("csrc_files", [])) != doc["csrc_sha256"]:
            msg = (f"{os.path.relpath(TRAFFIC_FILE, ROOT)} was taken on other kernel sources than this tree's (csrc hash differs): "
                   f"traffic not reported; re-run tools/pmc_traffic.sh")
            print("[bench] WARNING: " + msg, file=sys.stderr)
            return None, msg
    except Exception as e:
        msg = f"{os.path.relpath(TRAFFIC_FILE, ROOT)} unreadable ({e!r}): traffic not reported"
        print("[bench] WARNING: " + msg, file=sys.stderr)
        return None, msg
    e = tab.get(kernel_name)
    if not e:          # rocprofv3 prints every template argument (defaulted ones too): match on the instantiation's prefix
        key = kernel_name.rstrip(">").strip()
        hits = [v for k, v in tab.items() if k.rstrip(">").strip().startswith(key)]
        e = hits[0] if len(hits) == 1 else None
    if not e:
        msg = (f"dominant kernel '{kernel_name}' (batch {batch}) is absent from {os.path.relpath(TRAFFIC_FILE, ROOT)} "
               f"(has: {sorted(tab)[:6]}...): re-run tools/pmc_pass.sh")
        print("[bench] WARNING: " + msg, file=sys.stderr)
        return None, msg
    return int((2.0 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024), (
        f"{os.path.relpath(TRAFFIC_FILE, ROOT)}: FETCH_SIZE {e['FETCH_SIZE']:.0f} KB x2 (gfx950 correction) + WRITE_SIZE "
        f"{e['WRITE_SIZE']:.0f} KB, averaged over {e['n_FETCH_SIZE']} launches of this kernel in an eager pass (separate --pmc runs)")


def worker_leg(n_clients=16, n_requests=128, lone_requests=24, keep_timing=False):
    """Through the reference's caller shape: ONE consumer thread takes jobs from a bounded queue and blocks in
    ``worker.run_job`` (backends/worker_pool.py:294-341; here tools/minipool.MiniPool, replayed against a recording of the real
    pool by the CPU tests).  Everything ``run_job`` does is inside the measurement: tokenise + CLIP encode, noise draw + H2D,
    the sampler, RGB8 D2H, PNG.  (a) one closed-loop client: run_job latency; (b) `n_clients` closed-loop clients: the queue
    behind the running job is drained into batched passes (HipLcmWorker._drain)."""
    import statistics
    import threading
    from types import SimpleNamespace
    from tools import minipool
    from sdlcm_amd.backends.worker_factory import create_hip_worker
    old = {k: os.environ.get(k) for k in ("MODEL", "MODEL_ROOT")}
    os.environ["MODEL"] = "synthetic"
    os.environ.setdefault("MODEL_ROOT", "/nonexistent")
    w = create_hip_worker(worker_id=0)
    pool = minipool.MiniPool(lambda worker_id: w, {"m": "synthetic"}, "m", queue_max=64)
    w.bind_queue(pool.q)

    def req(i):
        return SimpleNamespace(prompt=f"benchmark prompt number {i} a lighthouse at dusk", size="512x512", num_inference_steps=4,
                               guidance_scale=1.0, seed=1000 + i, style_lora=None)

    def closed_loop(clients, total):
        lat, lock, nxt = [], threading.Lock(), [0]

        def client():
            while True:
                with lock:
                    i = nxt[0]
                    nxt[0] += 1
                if i >= total:
                    return
                t0 = time.perf_counter()
                png, seed = pool.submit_job(minipool.GenerationJob(req=req(i))).result(timeout=600)
                assert png[:8] == b"\x89PNG\r\n\x1a\n" and seed == 1000 + i
                with lock:
                    lat.append(time.perf_counter() - t0)
        th = [threading.Thread(target=client) for _ in range(clients)]
        t0 = time.perf_counter()
        [t.start() for t in th]
        [t.join() for t in th]
        return time.perf_counter() - t0, sorted(lat)

    try:
        # steady state: every (lane, batch size) plan tuned and captured before anything is timed (a first-use capture is
        # hundreds of milliseconds and would land in some request's latency)
        eng, key = w._engine, w._job_key(req(0))
        for lane in range(eng.n_lanes):
            for bsz in eng.batch_sizes:
                eng.run_batch(key, [w._prepare(req(i), key) for i in range(bsz)], lane)
        closed_loop(1, 3)
        closed_loop(n_clients, 3 * n_clients)
        dt1, lat1 = closed_loop(1, lone_requests)
        nb0 = len(w._engine.batcher.batches)
        dtn, latn = closed_loop(n_clients, n_requests)
        sizes = w._engine.batcher.batches[nb0:]
        extra = {"timing": list(w._engine.timing or [])} if keep_timing else {}
        return {**extra, "workload": f"run_job through a single-consumer pool-shaped loop (tools/minipool.py), 512x512 4 steps, PNG included; "
                            f"synthetic weights; {n_clients} closed-loop clients / 1 client",
                "images_per_s": round(n_requests / dtn, 2), "clients": n_clients, "requests": n_requests,
                "latency_p50_ms": round(statistics.median(latn) * 1e3, 2), "latency_p95_ms": round(latn[int(0.95 * (len(latn) - 1))] * 1e3, 2),
                "passes": len(sizes), "mean_batch": round(sum(sizes) / max(1, len(sizes)), 2),
                "batch_histogram": {str(k): sizes.count(k) for k in sorted(set(sizes))},
                "lone_client": {"images_per_s": round(lone_requests / dt1, 2), "run_job_p50_ms": round(statistics.median(lat1) * 1e3, 2),
                                "run_job_p95_ms": round(lat1[int(0.95 * (len(lat1) - 1))] * 1e3, 2)}}
    finally:
        w.bind_queue(None)
        pool._worker = None
        pool.shutdown()
        w.close()
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def spawn_command(gpus, argv, env):
    """`python bench.py --gpus N` run by hand (no torchrun environment): the command that launches N fresh ranks, one per
    GPU.  None when this process already IS a rank (RANK / WORLD_SIZE set by torch.distributed.run) or N == 1.  A
    WORLD_SIZE that disagrees with --gpus is an error, not something to paper over."""
    ws = env.get("WORLD_SIZE")
    if ws is not None:
        if int(ws) != int(gpus):
            raise SystemExit(f"bench.py: --gpus {gpus} but WORLD_SIZE={ws}: launch with --nproc-per-node {gpus}")
        return None
    if int(gpus) <= 1:
        return None
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(gpus)}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def claim_stdout():
    """Keep the process's real stdout for the ONE JSON line and point fd 1 at stderr for everything else (Python prints and
    C-level writes alike: RCCL's version banner, library chatter).  -> the saved descriptor for emit_line."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    return saved


def emit_line(saved_fd, line):
    """Write the JSON line (None: nothing -- ranks other than 0) to the descriptor claim_stdout() saved, and release it."""
    sys.stdout.flush()
    if line is not None:
        os.write(saved_fd, (json.dumps(line) + "\n").encode())
    os.close(saved_fd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1, help="images per GPU per step (BASELINE configs[1]: 1)")
    ap.add_argument("--size", type=int, default=None, help="image side (default 512; 1024 for --model sdxl)")
    ap.add_argument("--lcm-steps", type=int, default=None, help="sampler steps (default 4; 30 for --model sdxl)")
    ap.add_argument("--model", choices=("sd15", "sdxl"), default="sd15",
                    help="sd15 = BASELINE configs[1..3]; sdxl = configs[4] (SDXL-base 1024x1024, 30 steps, guidance 1.0, no CFG)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--no-roofline", action="store_true",
                    help="skip the roofline leg (profiling runs: the kernel trace then holds graph replays only, tools/pass_breakdown.py)")
    args = ap.parse_args()

    # N > 1 without a torchrun environment: become the launcher.  The parent never touches the GPU (no torch import, no
    # library call before this point); every rank is a fresh child process.
    cmd = spawn_command(args.gpus, sys.argv[1:], os.environ)
    if cmd is not None:
        import subprocess
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        raise SystemExit(subprocess.run(cmd, env=env).returncode)

    # The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a five-line version banner to stdout when
    # its first communicator comes up -- on every rank): keep the real stdout aside for the line and send everything else that
    # lands on fd 1 to stderr.
    real_stdout = claim_stdout()

    import numpy as np
    import torch
    import sdlcm_amd  # noqa: F401
    from sdlcm_amd import weights
    from sdlcm_amd.pipeline import LcmHipPipeline, draw_noise, guidance_scale_embedding

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # rehearsal only (one-GPU box): LCM_BENCH_BACKEND=gloo LCM_BENCH_ONE_DEVICE=1 runs N ranks on cuda:0 with CPU
    # collectives, exercising everything of the N>1 path except RCCL itself
    one_dev = os.environ.get("LCM_BENCH_ONE_DEVICE", "0") == "1"
    backend = os.environ.get("LCM_BENCH_BACKEND", "gloo" if one_dev else "nccl")      # RCCL refuses two ranks on one device
    if one_dev:
        local = 0
    # LCM_BENCH_FORCE_DIST=1: initialise the process group at world size 1 too, so that the RCCL code path of the N > 1 leg
    # (init_process_group("nccl", device_id=...), GPU-tensor broadcast / all_gather / all_reduce) loads and executes on a
    # one-GPU box -- the rehearsal of everything but the xGMI hop itself
    force_dist = os.environ.get("LCM_BENCH_FORCE_DIST", "0") == "1"
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    dev = f"cuda:{local}"
    if args.model == "sdxl":
        from sdlcm_amd.config import SDXL_UNET, unet_config, vae_config
        ucfg, vcfg = unet_config(SDXL_UNET), vae_config(dict(scaling_factor=0.13025, sample_size=1024, force_upcast=True))
        pipe = LcmHipPipeline(weights.synthetic_state_dict(weights.unet_param_spec(ucfg), 0),
                              weights.synthetic_state_dict(weights.vae_param_spec(vcfg), 1), ucfg, vcfg, device=dev)
    else:
        pipe = LcmHipPipeline(weights.synthetic_unet(), weights.synthetic_vae(), device=dev)
    B = args.batch
    S = args.size or (1024 if args.model == "sdxl" else 512)
    n = args.lcm_steps or (30 if args.model == "sdxl" else 4)
    h = w = S // 8
    D = pipe.unet.ctx_dim

    bcast_ms = {}
    encoders = {}
    if dist is not None:
        from sdlcm_amd.distributed import comm_stream       # collectives only ever run on that module's own never-captured stream

    def prompt_embeddings(pl, count):
        """What north_star describes: the rank that owns the prompt encoder (rank 0) runs it -- ClipTextHip on the HIP kernels,
        synthetic CLIP weights of the UNet's context width, hashed token ids of fixed prompts -- for ALL `count` requests of
        the job; the other ranks receive the result by broadcast."""
        from sdlcm_amd.clip import CLIP_BIGG, CLIP_L, ClipTextHip, HashTokenizer, synthetic_clip
        enc = encoders.get(id(pl))
        if enc is None:
            Dx = pl.unet.ctx_dim
            cfgs = [(CLIP_L, 2), (CLIP_BIGG, 3)] if Dx == 2048 else \
                   [(dict(CLIP_L, hidden_size=Dx, num_attention_heads=Dx // 64, intermediate_size=4 * Dx), 2)]
            enc = encoders[id(pl)] = [ClipTextHip(synthetic_clip(c, seed=sd), c, device=pl.device) for c, sd in cfgs]
        prompts = [f"benchmark prompt number {i} a lighthouse at dusk" for i in range(count)]
        outs = []
        for e in enc:
            ids = HashTokenizer(e.cfg["vocab_size"])(prompts)
            outs.append(e.forward(ids, output="penultimate") if len(enc) > 1 else e.forward(ids))
        return torch.cat(outs, dim=-1) if len(outs) > 1 else outs[0]

    def prime(Bx, lane=0, pl=None, hh=None, ww=None, nn=None, cfg=None):
        """Fill a plan's resident inputs: embeddings from rank 0's prompt encoder, broadcast over RCCL; per-request noise.
        cfg: a guidance scale > 1 = classifier-free guidance (UNet batch 2 Bx: rows [0, Bx) the negative conditioning = zeros,
        the reference's force_zeros_for_empty_prompt, rows [Bx, 2 Bx) the prompt)."""
        pl = pl or pipe
        hh, ww, nn = hh or h, ww or w, nn or n
        gd = float(cfg) if cfg else 1.0
        P = pl.plan(Bx, hh, ww, nn, do_cfg=bool(cfg), guidance=gd if cfg else None, lane=lane)
        with torch.cuda.stream(P.lane.stream):
            allpe = prompt_embeddings(pl, world * Bx) if rank == 0 else None
            if dist is not None:
                from sdlcm_amd.distributed import broadcast_embeddings
                P.lane.stream.synchronize()
                # Collectives run on a stream of their own, never on a lane's: ProcessGroupNCCL's watchdog thread polls the
                # events of recent work, and on HIP an event query fails ("operation not permitted on an event last recorded
                # in a capturing stream") once the stream it was recorded on is being captured -- which the lane's stream is,
                # a few seconds later (seen once at world size 1: the process group's watchdog took the process down)
                with comm_stream(dev if backend == "nccl" else "cpu"):
                    for rep in range(2):              # first call sets the communicator up; the second is the exchange itself
                        dist.barrier()
                        torch.cuda.synchronize()
                        tb = time.perf_counter()
                        if backend == "nccl":         # the one exchange step: 118 KB per prompt over xGMI
                            got = broadcast_embeddings(allpe, world * Bx, 77, pl.unet.ctx_dim, dev)
                        else:
                            got = broadcast_embeddings(allpe.cpu() if allpe is not None else None, world * Bx, 77, pl.unet.ctx_dim, "cpu").to(dev)
                        torch.cuda.synchronize()
                        bcast_ms[Bx] = (time.perf_counter() - tb) * 1e3
                allpe = got
            P.ehs[(P.UB - Bx) * 77:].copy_(allpe[rank * Bx:(rank + 1) * Bx].reshape(Bx * 77, pl.unet.ctx_dim))
            if pl.unet.has_added:      # SDXL: pooled text embedding + size/crop ids
                from sdlcm_amd.pipeline import sinusoid_host
                pooled = torch.randn(Bx, pl.unet.added_dim - 6 * 256, generator=torch.Generator().manual_seed(2))
                tid = torch.from_numpy(sinusoid_host(np.array([8 * hh, 8 * ww, 0, 0, 8 * hh, 8 * ww] * Bx, np.float32), 256)).reshape(Bx, -1)
                P.add_in[P.UB - Bx:].copy_(torch.cat([pooled, tid], 1).half())
                if cfg:
                    P.add_in[:Bx].copy_(torch.cat([torch.zeros_like(pooled), tid], 1).half())
            for b in range(Bx):
                l0, extra = draw_noise(1000 + rank * Bx + b, hh, ww, nn - 1)
                P.lat0[b].copy_(l0[0])
                for i, e in enumerate(extra):
                    P.noise[i, b].copy_(e[0])
            P.wemb.copy_(torch.from_numpy(guidance_scale_embedding(np.zeros(Bx, np.float32), P.wemb.shape[1])).half())
            pl.tune(P)                                 # per-shape launch autotune (once)
            pl._enqueue(P, gd)                         # eager warm-up: allocates scratch
            P.lane.stream.synchronize()
            from sdlcm_amd import ops
            g = ops.Graph()
            with g:
                pl._enqueue(P, gd)
            P.graph = g
        return P

    def timed_lanes(Ps, K, W):
        """K passes per lane with one pass of every lane in flight at a time (the serving mode at low load, DESIGN.md
        section 6): every lane replays its own graph on its own stream; -> (seconds, passes)."""
        for _ in range(W):
            for P in Ps:
                with torch.cuda.stream(P.lane.stream):
                    P.graph.launch()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            for P in Ps:
                with torch.cuda.stream(P.lane.stream):
                    P.graph.launch()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, K * len(Ps)

    def barrier():
        if dist is not None:
            with comm_stream(dev if backend == "nccl" else "cpu"):      # see prime(): never on a stream that gets captured
                dist.barrier()

    def timed(P, K, W):
        with torch.cuda.stream(P.lane.stream):
            for _ in range(W):
                P.graph.launch()
            torch.cuda.synchronize()
            barrier()
            torch.cuda.synchronize()
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
            t0 = time.perf_counter()
            evs[0].record()
            for i in range(K):
                P.graph.launch()
                evs[i + 1].record()
            torch.cuda.synchronize()
            barrier()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        lat = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(K))
        return dt, lat[len(lat) // 2]

    P = prime(B)
    dt, p50 = timed(P, args.steps, args.warmup)
    per_rank = [round(B * args.steps / dt, 3)]
    if dist is not None:
        with comm_stream(dev if backend == "nccl" else "cpu"):
            t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            allt = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(allt, t)
            per_rank = [round(B * args.steps / float(x.item()), 3) for x in allt]
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
    images = world * B * args.steps
    line = {
        "metric": f"{S}x{S} {n}-step LCM images/sec", "value": round(images / dt, 3), "unit": "images/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "p50_latency_ms": round(p50, 3), "per_rank_images_per_s": per_rank,
        "config": {"workload": f"{'SDXL-base' if args.model == 'sdxl' else 'SD1.5'} LCM {S}x{S}, {n} steps, guidance 1.0 (no CFG), "
                               f"batch {B} per GPU, fp16 (fp32 accumulate), hipGraph replay; seeded synthetic "
                               f"{'SDXL' if args.model == 'sdxl' else 'SD1.5'}-architecture weights",
                   "batch_per_gpu": B, "global_batch": world * B, "image": f"{S}x{S}", "lcm_steps": n,
                   "parallelism": f"independent requests x{world} (RCCL broadcast of prompt embeddings only)"},
    }
    if dist is not None:
        line["exchange"] = {"what": f"broadcast of [{world * B},77,{D}] fp16 prompt embeddings (rank 0's ClipTextHip output) before the "
                                    f"timed region; all_gather / all_reduce(MAX) of the per-rank times",
                            "backend": "rccl" if backend == "nccl" else backend, "ms": round(bcast_ms.get(B, 0.0), 3),
                            "world_size": world}
        # one real sharded generation through the same exchange code (distributed.run_sharded): every rank generates its shard
        # of 2 * world requests with pipe.generate, RGB8 gathered on rank 0 (outside the timed region; a functional check)
        from sdlcm_amd.distributed import run_sharded
        cdev = torch.device(dev if backend == "nccl" else "cpu")
        nreq = 2 * world
        pe_all = prompt_embeddings(pipe, nreq) if rank == 0 else None
        if pe_all is not None and backend != "nccl":
            pe_all = pe_all.cpu()

        def _gen(pe, seeds):
            out = pipe.generate(pe, seeds, 64, 64, 2, 1.0)
            return torch.from_numpy(out["rgb"]).to(cdev)
        full = run_sharded(_gen, pe_all, [5000 + i for i in range(nreq)], cdev, gather_to=0)
        if rank == 0:
            line["exchange"]["sharded_check"] = {"requests": nreq, "gathered": list(full.shape), "ok": bool(full.shape[0] == nreq and full.any())}
    if world > 1 and not args.no_extra and B == 1 and args.model == "sd15":
        # BASELINE configs[2] itself: 8 requests per GPU in one batched pass on EVERY rank (64 requests on 8 GPUs); `value`
        # stays the batch-1-per-GPU workload so that the N = 1 line is comparable
        P8 = prime(8)
        k8 = max(3, args.steps // 4)
        dt8, _ = timed(P8, k8, 1)
        with comm_stream(dev if backend == "nccl" else "cpu"):
            t8 = torch.tensor([dt8], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            all8 = [torch.zeros_like(t8) for _ in range(world)]
            dist.all_gather(all8, t8)
            dist.all_reduce(t8, op=dist.ReduceOp.MAX)
            torch.cuda.synchronize()
        line["extra_batch8"] = {"images_per_s": round(world * 8 * k8 / float(t8.item()), 2), "ms_per_step": round(float(t8.item()) / k8 * 1e3, 2),
                                "per_rank_images_per_s": [round(8 * k8 / float(x.item()), 2) for x in all8],
                                "pipeline_tflops": round(5.74e12 * world * 8 / (float(t8.item()) / k8) / 1e12, 1),
                                "global_batch": world * 8, "exchange_ms": round(bcast_ms.get(8, 0.0), 3),
                                "workload": f"BASELINE configs[2]: {world * 8} requests, 8 per GPU in one batched pass on each of {world} GPUs "
                                            f"(aggregate over ranks, max-over-ranks time)"}
    if rank == 0 and world == 1 and not args.no_roofline:
        line["roofline"] = roofline_leg(pipe, P, 1.0, dt / args.steps * 1e3)
        if args.model == "sdxl":
            fl_img = (6.761e12 * n + 10.470e12) * (S * S) / (1024 * 1024)
        else:
            fl_img = 5.74e12 * (S * S) / (512 * 512) * n / 4
        line["pipeline_tflops"] = round(fl_img * B / (dt / args.steps) / 1e12, 1)
        if not args.no_extra and B == 1 and args.model == "sd15":
            P8 = prime(8)
            dt8, p508 = timed(P8, max(3, args.steps // 4), 1)
            k8 = max(3, args.steps // 4)
            line["extra_batch8"] = {"images_per_s": round(8 * k8 / dt8, 2), "ms_per_step": round(dt8 / k8 * 1e3, 2),
                                    "pipeline_tflops": round(fl_img * 8 / (dt8 / k8) / 1e12, 1),
                                    "workload": "same, batch 8 per GPU (BASELINE configs[2] per-GPU shard)",
                                    "roofline": roofline_leg(pipe, P8, 1.0, dt8 / k8 * 1e3)}
            # the same batch-8 pass with TWO in flight (16 requests on the GPU, two lanes): the saturated serving mode of the
            # worker's micro-batcher (backends/batching.py: a full batch waiting behind a running pass goes to the second lane)
            P8b = prime(8, lane=1)
            dtl8, npass8 = timed_lanes([P8, P8b], k8, 1)
            line["extra_batch8_two_lanes"] = {"images_per_s": round(8 * npass8 / dtl8, 2), "ms_per_pass_per_lane": round(dtl8 / k8 * 1e3, 2),
                                              "workload": "two batch-8 passes in flight on two lanes (16 requests on one GPU); never `value`"}
        if not args.no_extra and B == 1 and args.model == "sd15":
            P1 = prime(1, lane=1)
            dtl, npass = timed_lanes([P, P1], args.steps, 2)
            line["extra_two_lanes"] = {"images_per_s": round(npass / dtl, 2), "ms_per_image_per_lane": round(dtl / args.steps * 1e3, 2),
                                       "workload": "same batch-1 requests, two in flight on two lanes of the pipeline (own stream / scratch / "
                                                   "graph / split-K workspace, shared weights); never `value`"}
        if not args.no_extra and B == 1 and args.model == "sd15" and S == 512:
            # BASELINE configs[3]: 768x768, 8 steps, batch 8 on one GPU (3 timed passes)
            P3 = prime(8, hh=96, ww=96, nn=8)
            dt3, _ = timed(P3, 3, 1)
            line["extra_768_b8"] = {"images_per_s": round(8 * 3 / dt3, 3), "ms_per_step": round(dt3 / 3 * 1e3, 2),
                                    "pipeline_tflops": round(22.95e12 * 8 / (dt3 / 3) / 1e12, 1),
                                    "workload": "SD1.5 LCM 768x768, 8 steps, batch 8 (BASELINE configs[3]; 22.95 TFLOP per image, SURVEY 8d)"}
            del P3
        if not args.no_extra and B == 1 and args.model == "sd15" and S == 512 and os.environ.get("LCM_BENCH_WORKER", "1") != "0":
            line["extra_worker"] = worker_leg()
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(host_threads())
        if not args.no_extra and B == 1 and args.model == "sd15" and S == 512 and os.environ.get("LCM_BENCH_SDXL", "1") != "0":
            # BASELINE configs[4]: SDXL-base architecture, 1024x1024, 30 steps, batch 1, guidance 1.0 (no CFG), 2 timed passes.
            # Its own pipeline (2.6 G synthetic parameters); the SD1.5 one is released first.
            pipe.close()
            encoders.clear()
            import gc
            gc.collect(); torch.cuda.empty_cache()
            from sdlcm_amd.config import SDXL_UNET, unet_config, vae_config
            ucfg, vcfg = unet_config(SDXL_UNET), vae_config(dict(scaling_factor=0.13025, sample_size=1024, force_upcast=True))
            xl = LcmHipPipeline(weights.synthetic_state_dict(weights.unet_param_spec(ucfg), 0),
                                weights.synthetic_state_dict(weights.vae_param_spec(vcfg), 1), ucfg, vcfg, device=dev)
            Px = prime(1, pl=xl, hh=128, ww=128, nn=30)
            dtx, _ = timed(Px, 2, 1)
            flx = 6.761e12 * 30 + 10.470e12
            line["extra_sdxl"] = {"images_per_s": round(2 / dtx, 4), "ms_per_step": round(dtx / 2 * 1e3, 1),
                                  "pipeline_tflops": round(flx / (dtx / 2) / 1e12, 1),
                                  "workload": "SDXL-base architecture 1024x1024, 30 steps, batch 1, guidance 1.0 (no CFG: UNet batch 1), fp16 "
                                              "VAE with residual-stream rescaling (BASELINE configs[4]; 213 TFLOP per image, SURVEY 8d)"}
            if os.environ.get("LCM_BENCH_SDXL_CFG", "1") != "0":
                # the reference's own SDXL mode default: guidance 7.5 (modes.yaml.example:48-54) => classifier-free guidance,
                # every UNet forward runs on [negative | prompt] = batch 2; the VAE decodes batch 1
                Pc = prime(1, pl=xl, hh=128, ww=128, nn=30, cfg=7.5)
                dtc, _ = timed(Pc, 2, 1)
                flc = 2 * 6.761e12 * 30 + 10.470e12
                line["extra_sdxl_cfg"] = {"images_per_s": round(2 / dtc, 4), "ms_per_step": round(dtc / 2 * 1e3, 1),
                                          "pipeline_tflops": round(flc / (dtc / 2) / 1e12, 1),
                                          "workload": "SDXL-base architecture 1024x1024, 30 steps, batch 1, guidance 7.5 (classifier-free guidance: "
                                                      "UNet batch 2; the reference's sdxl mode default, modes.yaml.example:48-54); 416 TFLOP per image"}
            xl.close()
    emit_line(real_stdout, line if rank == 0 else None)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
