"""Parity of the assembled HIP path against the CPU oracle (oracle/) on identical seeded inputs.

Tolerance: BASELINE.json north_star -- per-pixel |delta| < 1e-2 on the decoded image in [0,1]
(fp16 MI355X path vs fp32 CPU restatement).  Weights are the seeded synthetic SD1.5-architecture
weights (no checkpoint ships with the reference), stored as fp16 and shared by both sides.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def state():
    from sdlcm_amd import weights
    from sdlcm_amd.pipeline import LcmHipPipeline
    from oracle.pipeline import LCMPipelineOracle
    usd, vsd = weights.synthetic_unet(), weights.synthetic_vae()
    hip = LcmHipPipeline(usd, vsd, device="cuda:0")
    ora = LCMPipelineOracle(usd, vsd)
    return dict(hip=hip, ora=ora)


def _embeds(B, seed=5):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B, 77, 768, generator=g).to(torch.float16)


def _report(tag, got, ref):
    err = np.abs(got - ref)
    print(f"[parity] {tag}: max|d|={err.max():.4g} mean|d|={err.mean():.4g} ref_rms={np.sqrt((ref ** 2).mean()):.4g}")
    return err


def test_unet_forward_parity(state):
    from sdlcm_amd.pipeline import guidance_scale_embedding
    hip, ora = state["hip"], state["ora"]
    B, h, w, t = 2, 16, 16, 759
    g = torch.Generator().manual_seed(11)
    lat = torch.randn(B, 4, h, w, generator=g)
    pe = _embeds(B)
    wemb = torch.from_numpy(guidance_scale_embedding(np.zeros(B, np.float32), 256))
    ora.unet.taps = {}
    ref = ora.unet.forward(lat, t, pe.float(), wemb).numpy()
    ref_taps, ora.unet.taps = ora.unet.taps, None
    u = hip.unet
    taps = {}
    with torch.cuda.stream(hip.stream):
        ehs = pe.reshape(B * 77, 768).to(hip.device)
        kv = u.encode_context(ehs, B)
        eps = torch.zeros(B, h, w, 4, dtype=torch.float32, device=hip.device)
        u.forward(lat.to(hip.device), t, kv, wemb.to(hip.device, torch.float16), B, h, w, eps, taps=taps)
        hip.stream.synchronize()
    for k in ("conv_in", "down_blocks.0.resnets.0", "down_blocks.0.attentions.0", "down_blocks.2.attentions.1",
              "mid_block.resnets.1", "up_blocks.1.attentions.2", "up_blocks.3.attentions.2"):
        e = _report(k, taps[k].numpy(), ref_taps[k].numpy())
        assert e.max() < 0.05 * max(1.0, np.abs(ref_taps[k].numpy()).max()), k
    got = eps.cpu().numpy().transpose(0, 3, 1, 2)
    e = _report("unet eps", got, ref)
    assert e.max() < 2e-2


def test_time_embedding_of_all_steps_equals_per_step(state):
    """UNetHip.time_embed_all (the sampler's form: every step's TimestepEmbedding + guidance embedding + all time_emb_proj in
    one launch per layer, rows step-major) is bit-identical to time_embed called per step, for batch sizes on both sides of
    the kernel's 16-row block."""
    from sdlcm_amd.pipeline import guidance_scale_embedding
    hip = state["hip"]
    u = hip.unet
    ts = [999, 759, 499, 259, 19]
    for B in (1, 3, 8):
        wemb = torch.from_numpy(guidance_scale_embedding(np.linspace(0.0, 7.0, B).astype(np.float32), 256)).to(hip.device, torch.float16)
        with torch.cuda.stream(hip.stream):
            all_rows = u.time_embed_all(ts, wemb, B).clone()
            per_step = [u.time_embed(t, wemb, B).clone() for t in ts]
            hip.stream.synchronize()
        assert all_rows.shape == (len(ts) * B, u.temb_total)
        for i in range(len(ts)):
            assert torch.equal(all_rows[i * B:(i + 1) * B], per_step[i]), (B, i)
        assert not torch.equal(per_step[0], per_step[1])


def test_vae_decode_parity(state):
    hip, ora = state["hip"], state["ora"]
    B, h, w = 2, 16, 16
    lat = torch.randn(B, 4, h, w, generator=torch.Generator().manual_seed(3)) * 0.9
    ref = ora.vae.decode(lat).numpy()
    with torch.cuda.stream(hip.stream):
        rgb = torch.zeros(B, 8 * h, 8 * w, 3, dtype=torch.uint8, device=hip.device)
        img = torch.zeros(B, 8 * h, 8 * w, 3, dtype=torch.float32, device=hip.device)
        hip.vae.decode(lat.to(hip.device), B, h, w, rgb, img_f32=img)
        hip.stream.synchronize()
    got = img.cpu().numpy().transpose(0, 3, 1, 2)
    e = _report("vae image", np.clip(got / 2 + 0.5, 0, 1), np.clip(ref / 2 + 0.5, 0, 1))
    assert e.max() < 1e-2
    from oracle import glue
    u8 = glue.postprocess_u8(ref)
    assert np.abs(rgb.cpu().numpy().astype(int) - u8.astype(int)).max() <= 3


def test_vae_residual_stream_rescaling_matches_fp32_reference_where_fp16_overflows(state):
    """SDXL VAE ``force_upcast``: the reference decodes it in fp32 because the decoder's residual stream overflows fp16.
    The HIP decoder instead carries the stream scaled by 1/16 (exact reparametrisation, model.scale_vae_residual_stream).
    (1) plain weights: scaled == unscaled == oracle within tolerance; (2) weights whose residual stream reaches ~1e5:
    the unscaled fp16 decoder produces non-finite / wrong pixels, the scaled one still matches the fp32 oracle."""
    from sdlcm_amd.model import VAEDecoderHip
    from oracle.vae import VAEDecoderOracle
    hip, ora = state["hip"], state["ora"]
    B, h, w = 1, 16, 16
    lat = torch.randn(B, 4, h, w, generator=torch.Generator().manual_seed(11)) * 0.9
    base_sd = {k: v.clone() for k, v in ora.vae.sd.items()}
    for boost in (1.0, 6000.0):
        sd = dict(base_sd)
        if boost != 1.0:       # blow up the stream at its source; GroupNorms keep everything downstream of them O(1)
            sd["decoder.conv_in.weight"] = base_sd["decoder.conv_in.weight"] * boost
            sd["decoder.conv_in.bias"] = base_sd["decoder.conv_in.bias"] * boost
        ref = VAEDecoderOracle(sd, ora.vae.cfg).decode(lat).numpy()
        sd16 = {k: v.to(torch.float16) for k, v in sd.items()}
        imgs = {}
        for scale in (1.0, 1.0 / 16.0):
            cfg = dict(hip.vae.cfg, residual_scale=scale, force_upcast=False)
            with torch.cuda.stream(hip.stream):
                dec = VAEDecoderHip(sd16, cfg, device=hip.device)
                rgb = torch.zeros(B, 8 * h, 8 * w, 3, dtype=torch.uint8, device=hip.device)
                img = torch.zeros(B, 8 * h, 8 * w, 3, dtype=torch.float32, device=hip.device)
                dec.decode(lat.to(hip.device), B, h, w, rgb, img_f32=img)
                hip.stream.synchronize()
            imgs[scale] = np.clip(img.cpu().numpy().transpose(0, 3, 1, 2) / 2 + 0.5, 0, 1)
        r01 = np.clip(ref / 2 + 0.5, 0, 1)
        e_scaled = np.abs(imgs[1.0 / 16.0] - r01)
        e_plain = np.abs(np.nan_to_num(imgs[1.0], nan=9.0) - r01)
        print(f"boost {boost}: max err scaled {e_scaled.max():.4g}, unscaled {e_plain.max():.4g}")
        assert np.isfinite(imgs[1.0 / 16.0]).all() and e_scaled.max() < 1e-2
        if boost == 1.0:
            assert e_plain.max() < 1e-2
        else:
            assert e_plain.max() > 5e-2          # the un-rescaled fp16 decoder really does break on these weights


@pytest.mark.parametrize("h,w", [(48, 40), (32, 72), (56, 56)])
def test_vae_tiled_decode_parity(state, h, w):
    """vae.enable_tiling() (backends/cuda_worker.py:91): latents larger than sample_size/8 take diffusers' overlapping
    tile decode; a 256-px sample_size keeps the case small (2x2 ragged, 1x3, 3x3 tiles; latent sides stay multiples
    of 8 as every request size the worker accepts gives)."""
    hip, ora = state["hip"], state["ora"]
    B = 2
    lat = torch.randn(B, 4, h, w, generator=torch.Generator().manual_seed(h * 100 + w)) * 0.9
    old_o, old_h = ora.vae.cfg.get("sample_size", 512), hip.vae.cfg.get("sample_size", 512)
    ora.vae.cfg["sample_size"] = hip.vae.cfg["sample_size"] = 256
    try:
        ref = ora.vae.decode(lat).numpy()
        plain = ora.vae.decode(lat, use_tiling=False).numpy()
        with torch.cuda.stream(hip.stream):
            rgb = torch.zeros(B, 8 * h, 8 * w, 3, dtype=torch.uint8, device=hip.device)
            img = torch.zeros(B, 8 * h, 8 * w, 3, dtype=torch.float32, device=hip.device)
            hip.vae.decode(lat.to(hip.device), B, h, w, rgb, img_f32=img)
            hip.stream.synchronize()
    finally:
        ora.vae.cfg["sample_size"], hip.vae.cfg["sample_size"] = old_o, old_h
    assert np.abs(ref - plain).max() > 1e-2          # the tiled result really is a different image
    got = img.cpu().numpy().transpose(0, 3, 1, 2)
    e = _report(f"tiled vae {h}x{w}", np.clip(got / 2 + 0.5, 0, 1), np.clip(ref / 2 + 0.5, 0, 1))
    assert e.max() < 1e-2
    from oracle import glue
    assert np.abs(rgb.cpu().numpy().astype(int) - glue.postprocess_u8(ref).astype(int)).max() <= 3


@pytest.mark.parametrize("size,steps,seed", [(128, 4, 42), (64, 1, 7), (192, 2, 1234)])
def test_end_to_end_parity(state, size, steps, seed):
    hip, ora = state["hip"], state["ora"]
    pe = _embeds(1, seed=seed)
    ref = ora(pe.float(), size, size, steps, 1.0, seed)
    out = hip.generate(pe, [seed], size, size, steps, 1.0, want_float=True)
    _report(f"{size}px latents", out["latents"], ref["latents"])
    a = np.clip(out["image"].transpose(0, 3, 1, 2) / 2 + 0.5, 0, 1)
    b = np.clip(ref["image"] / 2 + 0.5, 0, 1)
    e = _report(f"{size}px {steps}-step image[0,1]", a, b)
    assert e.max() < 1e-2
    d8 = np.abs(out["rgb"].astype(int) - ref["image_u8"].astype(int))
    print(f"[parity] u8 max diff {d8.max()}, differing pixels {(d8 > 0).mean():.4f}")
    assert d8.max() <= 3


def test_non_square_768x512_parity(state):
    """BASELINE config 4 geometry (96-wide latents: levels 96/48/24/12, partial 16-wide patches at the 12-pixel level)."""
    hip, ora = state["hip"], state["ora"]
    pe = _embeds(1, seed=3)
    ref = ora(pe.float(), 768, 512, 2, 1.0, 11)
    out = hip.generate(pe, [11], 768, 512, 2, 1.0, want_float=True)
    assert out["rgb"].shape == (1, 512, 768, 3)
    _report("768x512 latents", out["latents"], ref["latents"])
    a = np.clip(out["image"].transpose(0, 3, 1, 2) / 2 + 0.5, 0, 1)
    b = np.clip(ref["image"] / 2 + 0.5, 0, 1)
    e = _report("768x512 2-step image[0,1]", a, b)
    assert e.max() < 1e-2


@pytest.mark.parametrize("width,height,steps", [(520, 392, 2), (360, 200, 1)])
def test_sizes_multiple_of_8_parity(state, width, height, steps):
    """Sizes the reference accepts (multiples of 8: backends/rknnlcm.py:380-381) that are not multiples of 64: odd latent
    sides at every UNet level (65x49 -> 33x25 -> 17x13 -> 9x7), upsamplers target the odd-sized skips, ragged attention
    sequences, GroupNorm statistics of images whose pixel count is not a multiple of 32, VAE attention with padded S."""
    hip, ora = state["hip"], state["ora"]
    pe = _embeds(1, seed=17)
    ref = ora(pe.float(), width, height, steps, 1.0, 99)
    out = hip.generate(pe, [99], width, height, steps, 1.0, want_float=True)
    assert out["rgb"].shape == (1, height, width, 3)
    _report(f"{width}x{height} latents", out["latents"], ref["latents"])
    a = np.clip(out["image"].transpose(0, 3, 1, 2) / 2 + 0.5, 0, 1)
    b = np.clip(ref["image"] / 2 + 0.5, 0, 1)
    e = _report(f"{width}x{height} {steps}-step image[0,1]", a, b)
    assert e.max() < 1e-2
    rep = hip.generate(pe, [99], width, height, steps, 1.0)
    assert np.array_equal(rep["rgb"], out["rgb"])
    with pytest.raises(Exception, match="divisible by 8"):
        hip.generate(pe, [99], 516, 392, 1, 1.0)


def test_graph_replay_equals_eager_and_is_deterministic(state):
    hip = state["hip"]
    pe = _embeds(1, seed=9)
    g1 = hip.generate(pe, [77], 128, 128, 4, 1.0)          # first graph use autotunes the launch plans
    eager = hip.generate(pe, [77], 128, 128, 4, 1.0, want_float=True)
    g2 = hip.generate(pe, [77], 128, 128, 4, 1.0)
    assert hip.plan(1, 16, 16, 4).graph is not None
    assert np.array_equal(g1["rgb"], g2["rgb"]) and np.array_equal(g1["latents"], g2["latents"])
    assert np.array_equal(g1["rgb"], eager["rgb"])
    other = hip.generate(pe, [78], 128, 128, 4, 1.0)
    assert not np.array_equal(other["rgb"], g1["rgb"])


@pytest.mark.parametrize("size,steps,B", [(128, 4, 3), (256, 2, 2), (512, 1, 2)])
def test_batched_requests_match_single_requests(state, size, steps, B):
    """A request inside a batch gets exactly the bytes it gets alone (the reference's same-seed contract,
    tests/test_sdxl_worker.py:171-198, extended to micro-batches): K partitions, reduce slabs and statistics slabs are
    keyed on the per-image shape.  512x512 at batch 2 also flips the VAE's 256^2 level from the separate GroupNorm-apply
    pass to the form fused into the conv's halo staging -- same bits."""
    hip = state["hip"]
    pe = _embeds(B, seed=21)
    seeds = [1000 + i for i in range(B)]
    batched = hip.generate(pe, seeds, size, size, steps, 1.0)
    for i, s in enumerate(seeds):
        one = hip.generate(pe[i:i + 1], [s], size, size, steps, 1.0)
        assert np.array_equal(one["latents"][0], batched["latents"][i]), f"latents of request {i} depend on the batch"
        assert np.array_equal(one["rgb"][0], batched["rgb"][i]), f"pixels of request {i} depend on the batch"


@pytest.mark.parametrize("W,H,B", [(640, 360, 2), (512, 768, 2)])
def test_batched_requests_match_single_requests_ui_sizes(state, W, H, B):
    """The same contract at the reference UI's non-square stock sizes (lcm-sr-ui/src/utils/constants.js:6-15), whose launch
    plans and per-image K partitions come from the shipped table (tools/extend_plans.py): 80x45 latents have odd UNet
    levels (45 -> 23 -> 12 -> 6) and a ragged 16-wide patch grid."""
    hip = state["hip"]
    pe = _embeds(B, seed=22)
    seeds = [1100 + i for i in range(B)]
    batched = hip.generate(pe, seeds, W, H, 1, 1.0)
    assert batched["rgb"].shape == (B, H, W, 3)
    for i, s in enumerate(seeds):
        one = hip.generate(pe[i:i + 1], [s], W, H, 1, 1.0)
        assert np.array_equal(one["latents"][0], batched["latents"][i]), f"latents of request {i} depend on the batch"
        assert np.array_equal(one["rgb"][0], batched["rgb"][i]), f"pixels of request {i} depend on the batch"


def test_two_lanes_in_flight_match_solo_runs(state):
    """Two requests in flight at once on two lanes of one pipeline (own stream, scratch, graphs and split-K workspace;
    shared weights): every result is bit-identical to the same request run alone, repeatedly, with the lanes racing."""
    import threading
    hip = state["hip"]
    pe = _embeds(4, seed=33)
    reqs = [(pe[i:i + 1], 2000 + i) for i in range(4)]
    solo = [hip.generate(e, [s], 256, 256, 4, 1.0) for e, s in reqs]
    assert hip.lane(1).stream is not hip.lane(0).stream and hip.lane(1).unet.w is hip.unet.w
    assert hip.lane(1).splitk_ws.data_ptr() != hip.lane(0).splitk_ws.data_ptr()
    got, errs = {}, []

    def run(lane):
        try:
            for rep in range(6):
                for k in range(4):
                    i = (k + lane * 2) % 4
                    out = hip.generate(reqs[i][0], [reqs[i][1]], 256, 256, 4, 1.0, lane=lane)
                    got[(lane, rep, i)] = (out["rgb"], out["latents"])
        except BaseException as e:      # noqa
            errs.append(e)
    th = [threading.Thread(target=run, args=(lane,)) for lane in (0, 1)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    assert len(got) == 48
    for (lane, rep, i), (rgb, lat) in got.items():
        assert np.array_equal(rgb, solo[i]["rgb"]) and np.array_equal(lat, solo[i]["latents"]), (lane, rep, i)


def test_tuning_record_hook_is_thread_local(state):
    """The autotuner's record hook only sees launches of the thread that opened it (round 2: a process-global hook collected the
    eager prompt-encoder launches of the other lane, and the tuner replayed them on that lane's live buffers -- in-place
    residual GEMMs accumulated).  While one thread records a sampler pass on lane 1, another encodes prompts eagerly on lane 0:
    no text-encoder shape (K = 768 with 77-row images) reaches the record list, the encoder outputs stay bit-identical, and
    the other thread's own hook state is untouched."""
    import threading
    from sdlcm_amd import ops
    from sdlcm_amd.clip import ClipTextHip, synthetic_clip
    hip = state["hip"]
    enc = ClipTextHip(synthetic_clip(), None, device=hip.device)
    ids = torch.randint(1, 49000, (2, 77), dtype=torch.int32)
    s0 = torch.cuda.Stream(device=hip.device)
    with torch.cuda.stream(s0):
        ref = enc.forward(ids).clone()
        s0.synchronize()
    stop, outs, seen, errs = threading.Event(), [], [], []

    def encode_loop():
        try:
            with torch.cuda.stream(s0):
                while not stop.is_set():
                    seen.append(ops._record_list())
                    outs.append(enc.forward(ids).clone())
                s0.synchronize()
        except BaseException as e:      # noqa
            errs.append(e)
    P = hip.plan(1, 16, 16, 2, lane=1)
    th = threading.Thread(target=encode_loop)
    with torch.cuda.stream(P.lane.stream):
        hip._enqueue(P, 1.0)                       # allocates this plan's scratch
        th.start()
        with ops.recording() as recs:
            for _ in range(3):
                hip._enqueue(P, 1.0)
        P.lane.stream.synchronize()
    stop.set()
    th.join()
    assert not errs, errs
    assert len(outs) >= 1 and all(r is None for r in seen)
    assert ops._record_list() is None
    keys = [r[0] for r in recs if r[0] is not None]
    assert keys and not any(k[0] == 0 and k[3] == 768 and k[1] % 77 == 0 and k[2] != hip.unet.kv_total for k in keys), "text-encoder launches leaked into the record"
    n_pass = len([r for r in recs])
    assert n_pass % 3 == 0                          # three identical passes, nothing else
    for o in outs:
        assert torch.equal(o, ref)


def test_graphs_of_earlier_plans_survive_larger_plans(state):
    """A captured hipGraph bakes raw pointers in.  Building and running a batch-8 plan (bigger GroupNorm workspaces, scratch,
    statistics buffers) after a batch-1 plan must not free or move anything the batch-1 graph still uses: the batch-1
    request replays to the same bytes afterwards (round 1 grew one shared GroupNorm workspace in place)."""
    hip = state["hip"]
    pe1, pe8 = _embeds(1, seed=51), _embeds(8, seed=52)
    first = hip.generate(pe1, [31], 256, 256, 2, 1.0)
    assert hip.plan(1, 32, 32, 2).graph is not None
    big = hip.generate(pe8, list(range(40, 48)), 256, 256, 2, 1.0)
    junk = [torch.full((1 << 22,), 7.0, device=hip.device) for _ in range(8)]      # would land in anything that was freed
    torch.cuda.synchronize()
    again = hip.generate(pe1, [31], 256, 256, 2, 1.0)
    del junk
    assert np.array_equal(again["rgb"], first["rgb"]) and np.array_equal(again["latents"], first["latents"])
    big2 = hip.generate(pe8, list(range(40, 48)), 256, 256, 2, 1.0)
    assert np.array_equal(big2["rgb"], big["rgb"])


def test_latents_blob_matches_oracle_pooling(state):
    from oracle import glue
    hip = state["hip"]
    out = hip.generate(_embeds(1), [5], 128, 128, 2, 1.0)
    blob = out["pool8"][:1].tobytes(order="C")
    assert len(blob) == 512
    ref = np.frombuffer(glue.latents_blob(out["latents"][:1]), np.float16)
    assert np.abs(np.frombuffer(blob, np.float16).astype(np.float32) - ref.astype(np.float32)).max() < 2e-2


@pytest.mark.parametrize("B,act", [(1, "quick_gelu"), (3, "quick_gelu"), (2, "gelu")])
def test_clip_text_encoder_parity(B, act):
    """Native CLIP text encoder (token gather, LN, fused-QKV GEMM, causal flash attention, quick-GELU MLP) against
    transformers.CLIPTextModel on the CPU in fp32, same (synthetic, CLIP-L sized) weights and token ids."""
    from sdlcm_amd.clip import CLIP_L, ClipTextHip, HashTokenizer, synthetic_clip
    from oracle.clip import clip_text_oracle
    cfg = dict(CLIP_L, hidden_act=act)
    sd = synthetic_clip(cfg)
    ids = HashTokenizer()(["a photo of an astronaut riding a horse on mars", "", "x " * 100][:B])
    ref = clip_text_oracle(sd, cfg, ids).numpy()
    enc = ClipTextHip(sd, cfg, device="cuda:0")
    got = enc.forward(ids).float().cpu().numpy()
    e = _report(f"clip last_hidden_state B={B} {act}", got, ref)
    assert e.max() < 2e-2 * max(1.0, np.abs(ref).max())


def test_clip_graph_replay_equals_eager_and_follows_new_prompts():
    """The SD1.5 prompt encoder replays a captured hipGraph per (B, 77): same bits as the eager kernel sequence, for the
    prompt it was captured with and for later prompts (token ids live in a static device buffer), per batch size; the
    returned tensor is the caller's own (a later encode must not overwrite it)."""
    from sdlcm_amd import clip as clipmod
    from sdlcm_amd.clip import CLIP_L, ClipTextHip, HashTokenizer, synthetic_clip
    cfg = dict(CLIP_L, num_hidden_layers=4)
    sd = synthetic_clip(cfg)
    tok = HashTokenizer()
    enc = ClipTextHip(sd, cfg, device="cuda:0")
    prompts = [["a lighthouse at dusk"], ["two red kites over a beach"], ["a lighthouse at dusk", "", "oil painting of a cat"], [""]]
    old = clipmod.CLIP_GRAPH
    try:
        clipmod.CLIP_GRAPH = False
        eager = [enc.forward(tok(p)).clone() for p in prompts]
        clipmod.CLIP_GRAPH = True
        torch.cuda.synchronize()
        with torch.cuda.stream(torch.cuda.Stream(device="cuda:0")):     # as a pipeline lane does; the default stream stays eager
            first = enc.forward(tok(prompts[0]))
            graphed = [first] + [enc.forward(tok(p)) for p in prompts[1:]]
        torch.cuda.synchronize()
    finally:
        clipmod.CLIP_GRAPH = old
    assert len(enc._graphs) == 2                                     # (1, 77) and (3, 77)
    for e, g in zip(eager, graphed):
        assert torch.equal(e, g)
    assert not torch.equal(graphed[0], graphed[1])


def test_sdxl_style_unet_parity():
    """The SDXL UNet family (row a16): 3 levels, no attention at level 0, transformer depth 1/2/3, 64-wide heads, Linear
    proj_in/out, "text_time" additional embedding, no time_cond_proj -- narrow widths, HIP executor vs oracle."""
    from sdlcm_amd import weights
    from sdlcm_amd.config import SDXL_UNET, unet_config
    from sdlcm_amd.model import UNetHip
    from oracle.unet import UNetOracle, timestep_sinusoid
    cfg = unet_config(dict(SDXL_UNET, block_out_channels=(64, 128, 256), attention_head_dim=(1, 2, 4), cross_attention_dim=128,
                           transformer_layers_per_block=(1, 2, 3), addition_time_embed_dim=32,
                           projection_class_embeddings_input_dim=64 + 6 * 32))
    sd = weights.synthetic_state_dict(weights.unet_param_spec(cfg), 0)
    B, h, w, t = 2, 16, 24, 499
    g = torch.Generator().manual_seed(4)
    lat = torch.randn(B, 4, h, w, generator=g)
    ehs = torch.randn(B, 77, 128, generator=g).half()
    pooled = torch.randn(B, 64, generator=g).half()
    time_ids = torch.tensor([[h * 8, w * 8, 0, 0, h * 8, w * 8]] * B, dtype=torch.float32)
    ora = UNetOracle(sd, cfg)
    ref = ora.forward(lat, t, ehs.float(), None, added=(pooled.float(), time_ids)).numpy()
    dev = torch.device("cuda:0")
    u = UNetHip(sd, cfg, dev)
    add_in = torch.cat([pooled.float(), timestep_sinusoid(time_ids.reshape(-1), 32).reshape(B, -1)], -1).half().to(dev)
    kv = u.encode_context(ehs.reshape(B * 77, 128).to(dev), B)
    aug = u.encode_added(add_in, B)
    eps = torch.zeros(B, h, w, 4, dtype=torch.float32, device=dev)
    u.forward(lat.to(dev), t, kv, None, B, h, w, eps, aug=aug)
    torch.cuda.synchronize()
    e = _report("sdxl-style unet eps", eps.cpu().numpy().transpose(0, 3, 1, 2), ref)
    assert e.max() < 2e-2


def test_clip_sdxl_outputs_parity():
    """SDXL text conditioning: hidden_states[-2] of both encoders and the projected pooled embedding of the second
    (CLIPTextModelWithProjection), vs transformers on the CPU; bigG-shaped encoder at reduced depth (8 layers)."""
    from sdlcm_amd.clip import CLIP_BIGG, CLIP_L, ClipTextHip, HashTokenizer, synthetic_clip
    from oracle.clip import clip_text_oracle_sdxl
    ids = HashTokenizer()(["a watercolor painting of a fox in the snow", ""])
    for cfg in (dict(CLIP_L, num_hidden_layers=4), dict(CLIP_BIGG, num_hidden_layers=8)):
        sd = synthetic_clip(cfg, seed=5)
        ref_h, ref_p = clip_text_oracle_sdxl(sd, cfg, ids)
        enc = ClipTextHip(sd, cfg, device="cuda:0")
        if ref_p is None:
            got_h = enc.forward(ids, output="penultimate")
        else:
            got_h, got_p = enc.forward(ids, output="penultimate", pooled=True)
            e = _report(f"clip text_embeds D={cfg['hidden_size']}", got_p.float().cpu().numpy(), ref_p.numpy())
            assert e.max() < 2e-2 * max(1.0, np.abs(ref_p.numpy()).max())
        e = _report(f"clip hidden_states[-2] D={cfg['hidden_size']}", got_h.float().cpu().numpy(), ref_h.numpy())
        assert e.max() < 2e-2 * max(1.0, np.abs(ref_h.numpy()).max())


@pytest.mark.parametrize("guidance", [1.0, 5.0])
def test_sdxl_style_pipeline_parity(guidance):
    """SDXL-family sampler end to end (text_time conditioning, classifier-free guidance captured in the graph,
    VAE scaling 0.13025) on narrow synthetic weights: HIP vs oracle, eager and graph replay."""
    from sdlcm_amd import weights
    from sdlcm_amd.config import SDXL_UNET, unet_config, vae_config
    from sdlcm_amd.pipeline import LcmHipPipeline
    from oracle.pipeline import LCMPipelineOracle
    ucfg = unet_config(dict(SDXL_UNET, block_out_channels=(64, 128, 256), attention_head_dim=(1, 2, 4), cross_attention_dim=128,
                            transformer_layers_per_block=(1, 2, 2), addition_time_embed_dim=32,
                            projection_class_embeddings_input_dim=64 + 6 * 32))
    vcfg = vae_config(dict(block_out_channels=(64, 64, 128, 128), scaling_factor=0.13025))
    usd = weights.synthetic_state_dict(weights.unet_param_spec(ucfg), 0)
    vsd = weights.synthetic_state_dict(weights.vae_param_spec(vcfg), 1)
    hip = LcmHipPipeline(usd, vsd, ucfg, vcfg, device="cuda:0")
    ora = LCMPipelineOracle(usd, vsd, ucfg, vcfg)
    g = torch.Generator().manual_seed(8)
    pe = torch.randn(1, 77, 128, generator=g).half()
    ne = torch.randn(1, 77, 128, generator=g).half()
    pooled = torch.randn(1, 64, generator=g).half()
    tids = torch.tensor([[128.0, 192.0, 0, 0, 128.0, 192.0]])
    kw = dict(added=(pooled, tids), negative_embeds=ne if guidance > 1 else None)
    ref = ora(pe.float(), 192, 128, 3, guidance, 21, negative_embeds=ne.float() if guidance > 1 else None, added=(pooled.float(), tids))
    out = hip.generate(pe, [21], 192, 128, 3, guidance, want_float=True, **kw)
    a = np.clip(out["image"].transpose(0, 3, 1, 2) / 2 + 0.5, 0, 1)
    b = np.clip(ref["image"] / 2 + 0.5, 0, 1)
    e = _report(f"sdxl-style pipeline g={guidance} image[0,1]", a, b)
    assert e.max() < 1e-2
    g1 = hip.generate(pe, [21], 192, 128, 3, guidance, **kw)
    g2 = hip.generate(pe, [21], 192, 128, 3, guidance, **kw)
    assert np.array_equal(g1["rgb"], g2["rgb"])
    assert np.array_equal(g1["rgb"], out["rgb"])      # graph (tuned launch plans) vs the earlier eager pass: plans never change bits
    hip.drop_plans()


def test_text_encoder_lora_merge_parity_and_restore():
    """Text-encoder LoRA entries of a style file (kohya ``lora_te_`` / ``lora_te2_`` and peft ``text_encoder.`` keys; q/k/v
    rows of the fused projection, out_proj, fc1, fc2) merged in place into the native CLIP encoder: output must match
    transformers' CLIP on explicitly merged weights, weight 0 must restore the base output bit for bit."""
    from sdlcm_amd.clip import CLIP_L, ClipTextHip, HashTokenizer, synthetic_clip
    from sdlcm_amd.lora import ClipLora, parse_te_lora
    from oracle.clip import clip_text_oracle
    cfg = dict(CLIP_L, num_hidden_layers=3)
    sd = synthetic_clip(cfg, seed=7)
    g = torch.Generator().manual_seed(41)
    targets = [(0, "self_attn.q_proj", "kohya"), (0, "self_attn.v_proj", "peft"), (1, "self_attn.out_proj", "kohya"),
               (1, "mlp.fc1", "peft"), (2, "mlp.fc2", "kohya"), (2, "self_attn.k_proj", "kohya")]
    raw, merged, r, alpha, wt = {}, dict(sd), 4, 2.0, 0.8
    for layer, sub, style in targets:
        name = f"encoder.layers.{layer}.{sub}.weight"
        W = sd[name]
        down = torch.randn(r, W.shape[1], generator=g) * (W.shape[1] ** -0.5)
        up = torch.randn(W.shape[0], r, generator=g) * 0.3
        path = f"text_model.encoder.layers.{layer}.{sub}"
        if style == "kohya":
            k = "lora_te_" + path.replace(".", "_")
            raw[k + ".lora_down.weight"], raw[k + ".lora_up.weight"], raw[k + ".alpha"] = down, up, torch.tensor(alpha)
            scale = alpha / r
        else:
            k = "text_encoder." + path
            raw[k + ".lora_A.weight"], raw[k + ".lora_B.weight"] = down, up
            scale = 1.0
        merged[name] = (W.float() + wt * ((up @ down) * scale).to(torch.float16).float()).to(torch.float16)
    raw["lora_te2_text_model_encoder_layers_0_mlp_fc1.lora_down.weight"] = torch.zeros(r, 8)      # second encoder: not ours
    raw["lora_unet_mid_block_attentions_0_proj_in.lora_down.weight"] = torch.zeros(r, 8)
    parsed, used = parse_te_lora(raw, 3, 0)
    assert len(parsed) == len(targets) and all("te2" not in k and "unet" not in k for k in used)
    ids = HashTokenizer()(["a papercut illustration of a fox", ""])
    enc = ClipTextHip(sd, cfg, device="cuda:0")
    base = enc.forward(ids).clone()
    lora = ClipLora(enc, raw, 0)
    assert len(lora.modules) == len(targets)
    lora.apply(wt)
    got = enc.forward(ids).float().cpu().numpy()
    ref = clip_text_oracle(merged, cfg, ids).numpy()
    e = _report("clip with text-encoder LoRA", got, ref)
    assert e.max() < 2e-2 * max(1.0, np.abs(ref).max())
    assert np.abs(got - base.float().cpu().numpy()).max() > 1e-2         # the merge really changed the encoder
    lora.apply(0.0)
    assert torch.equal(enc.forward(ids), base)


def test_lora_style_merge_parity_and_restore(state):
    """A synthetic LoRA over every supported target kind (fused q|k|v rows, stacked cross-attention K/V, GEGLU
    interleave, proj_in/out, out-projections, ff.net.2): the in-place merge W + weight * (BA) alpha/r must match the
    oracle run on explicitly merged weights; weight 0 must restore the base output bit for bit."""
    from sdlcm_amd import weights
    from sdlcm_amd.lora import LoraStyle
    from oracle.pipeline import LCMPipelineOracle
    hip, ora = state["hip"], state["ora"]
    usd = weights.synthetic_unet()
    g = torch.Generator().manual_seed(31)
    targets = ["down_blocks.0.attentions.0.proj_in", "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q",
               "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_v", "down_blocks.1.attentions.1.transformer_blocks.0.attn1.to_out.0",
               "down_blocks.1.attentions.1.transformer_blocks.0.attn2.to_q", "mid_block.attentions.0.transformer_blocks.0.attn2.to_k",
               "mid_block.attentions.0.transformer_blocks.0.attn2.to_v", "up_blocks.1.attentions.0.transformer_blocks.0.attn2.to_out.0",
               "up_blocks.2.attentions.1.transformer_blocks.0.ff.net.0.proj", "up_blocks.3.attentions.2.transformer_blocks.0.ff.net.2",
               "up_blocks.3.attentions.2.proj_out"]
    raw, r, alpha, wt = {}, 8, 4.0, 0.9
    merged = dict(usd)
    # LoCon entries: 3x3 resnet convs, the 1x1 shortcut, time_emb_proj, a stride-2 downsampler and a (phase-packed) upsampler
    conv_targets = ["down_blocks.0.resnets.1.conv1", "up_blocks.1.resnets.0.conv2", "up_blocks.2.resnets.0.conv_shortcut",
                    "mid_block.resnets.0.time_emb_proj", "down_blocks.1.downsamplers.0.conv", "up_blocks.1.upsamplers.0.conv",
                    "conv_in", "conv_out"]
    for m in conv_targets:
        W = usd[m + ".weight"]
        key = "lora_unet_" + m.replace(".", "_")
        if W.dim() == 4:
            kk = W.shape[-1]
            down = torch.randn(r, W.shape[1], kk, kk, generator=g) * ((W.shape[1] * kk * kk) ** -0.5)
            up = torch.randn(W.shape[0], r, 1, 1, generator=g) * 0.5
            delta = torch.einsum("or,rikl->oikl", up.reshape(W.shape[0], r), down) * (alpha / r)
        else:
            down = torch.randn(r, W.shape[1], generator=g) * (W.shape[1] ** -0.5)
            up = torch.randn(W.shape[0], r, generator=g) * 0.5
            delta = (up @ down) * (alpha / r)
        raw[key + ".lora_down.weight"], raw[key + ".lora_up.weight"], raw[key + ".alpha"] = down, up, torch.tensor(alpha)
        merged[m + ".weight"] = (W.float() + wt * delta.to(torch.float16).float()).to(torch.float16)
    for i, m in enumerate(targets):
        W = usd[m + ".weight"]
        out_f, in_f = W.shape[0], W.reshape(W.shape[0], -1).shape[1]
        down = torch.randn(r, in_f, generator=g) * (in_f ** -0.5)
        up = torch.randn(out_f, r, generator=g) * 0.5
        key = "lora_unet_" + m.replace(".", "_") if i % 2 == 0 else "unet." + m
        if i % 2 == 0:
            raw[key + ".lora_down.weight"], raw[key + ".lora_up.weight"], raw[key + ".alpha"] = down, up, torch.tensor(alpha)
            scale = alpha / r
        else:
            raw[key + ".lora_A.weight"], raw[key + ".lora_B.weight"] = down, up
            scale = 1.0
        delta = ((up @ down) * scale).to(torch.float16)
        merged[m + ".weight"] = (W.float().reshape(out_f, in_f) + wt * delta.float()).to(torch.float16).reshape(W.shape)
    pe = _embeds(1, seed=12)
    base = hip.generate(pe, [4], 128, 128, 2, 1.0, want_float=True)
    with torch.cuda.stream(hip.stream):
        style = LoraStyle(hip.unet, raw)
        assert len(style.modules) == len(targets) + len(conv_targets) and not style.skipped
        style.apply(wt)
    try:
        out = hip.generate(pe, [4], 128, 128, 2, 1.0, want_float=True)
        ref = LCMPipelineOracle(merged, weights.synthetic_vae())(pe.float(), 128, 128, 2, 1.0, 4)
        a = np.clip(out["image"].transpose(0, 3, 1, 2) / 2 + 0.5, 0, 1)
        b = np.clip(ref["image"] / 2 + 0.5, 0, 1)
        e = _report("lora-merged image[0,1]", a, b)
        assert e.max() < 1e-2
        assert np.abs(out["image"] - base["image"]).max() > 1e-3          # the style does change the picture
    finally:
        with torch.cuda.stream(hip.stream):
            style.apply(0.0)
    again = hip.generate(pe, [4], 128, 128, 2, 1.0, want_float=True)
    assert np.array_equal(again["rgb"], base["rgb"]) and np.array_equal(again["latents"], base["latents"])
