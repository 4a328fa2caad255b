"""CPU-side tests of the product: host glue against the golden vectors and the oracle, weight inventory,
packing, the C ABI surface (library loads and exports every symbol include/lcm_hip.h declares), and the
loud failure of the product path without a GPU."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "glue_golden.npz"))


def test_abi_exports_every_declared_symbol():
    import __graft_entry__
    __graft_entry__.build()
    from sdlcm_amd import lib
    L = lib.load()
    hdr = open(os.path.join(ROOT, "include", "lcm_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(lcm_[a-z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(L, name), f"liblcmhip.so does not export {name}"
    assert sorted(lib.EXPORTS) == declared
    assert L.lcm_version() >= 100


def test_abi_argument_validation_without_gpu():
    """Precondition checks run before any launch, so they are testable on the CPU box."""
    import ctypes as C
    from sdlcm_amd import lib
    L = lib.load()
    rc = L.lcm_gemm_f16(None, 0, None, 0, 0, None, None, None, 0, 0, None, 0, None, 0, 1, 64, 64, 0, 1.0, 1, 0, 0, 0, 0, None, 0, None, None)
    assert rc == -1 and b"null pointer" in L.lcm_last_error()
    buf = C.create_string_buffer(16)
    p = C.cast(buf, C.c_void_p)
    rc = L.lcm_gemm_f16(p, 8, None, 0, 0, p, None, None, 0, 0, None, 0, p, 8, 4, 64, 100, 0, 1.0, 1, 0, 0, 0, 0, None, 0, None, None)
    assert rc == -1 and b"multiple of 64" in L.lcm_last_error()
    rc = L.lcm_attention_f16(p, 8, p, 8, p, 8, p, 8, 1, 8, 4, 4, 48, 1.0, 0, None)
    assert rc == -1 and b"head_dim" in L.lcm_last_error()
    with pytest.raises(lib.LcmHipError):
        lib.check(rc, "attention")
    assert L.lcm_gemm_tile_config(262144, 128, 1) == 128128


def test_undersized_statistics_buffer_is_refused_before_any_launch():
    """The API hole behind round 2's GPU fault: the contraction entry points take the size of the statistics buffer and return
    LCM_EINVAL when the launch's slabs would not fit (checked before anything is enqueued, so it runs on the CPU box too)."""
    import ctypes as C
    from sdlcm_amd import lib
    L = lib.load()
    buf = C.create_string_buffer(1 << 16)
    p = C.cast(buf, C.c_void_p)
    sp = C.c_int(0)
    # GEMM [64, 64] of one image: 2 slabs x 64 channels x (sum, sumsq) fp32 = 1024 bytes
    rc = L.lcm_gemm_f16(p, 64, None, 0, 0, p, None, None, 0, 0, None, 0, p, 64, 64, 64, 64, 0, 1.0, 1, 0, 0, 0, 64, p, 1023, C.byref(sp), None)
    assert rc == -1 and b"statistics buffer too small" in L.lcm_last_error() and b"1024 bytes" in L.lcm_last_error()
    # 3x3 conv of one 16x16 image, 64 -> 64 channels: 8 slabs of 2x16 pixels = 4096 bytes
    rc = L.lcm_conv3x3_f16(p, p, None, None, 0, None, p, 1, 16, 16, 64, 64, 1, 0, p, 4095, C.byref(sp), None)
    assert rc == -1 and b"statistics buffer too small" in L.lcm_last_error() and b"4096 bytes" in L.lcm_last_error()
    rc = L.lcm_conv3x3_gn_f16(p, 64, None, 0, None, None, 0, p, None, None, 0, None, p, 1, 16, 16, 64, 0, p, 4095, C.byref(sp), None)
    assert rc == -1 and b"statistics buffer too small" in L.lcm_last_error()
    # a statistics pointer without the slab-count out parameter is an error too
    rc = L.lcm_gemm_f16(p, 64, None, 0, 0, p, None, None, 0, 0, None, 0, p, 64, 64, 64, 64, 0, 1.0, 1, 0, 0, 0, 64, p, 1 << 16, None, None)
    assert rc == -1 and b"slabs_per_image" in L.lcm_last_error()
    # the sizing function always covers the slabs of a launch (odd image sizes included)
    assert L.lcm_stats_bytes(64, 64, 64) >= 1024 and L.lcm_stats_bytes(45 * 25, 320, 45 * 25) >= (23 * 3) * 320 * 8
    assert L.lcm_stats_bytes(8 * 4096, 320, 4096) == 8 * L.lcm_stats_bytes(4096, 320, 4096)


def test_product_glue_matches_reference_vectors():
    from sdlcm_amd.pipeline import draw_noise, guidance_scale_embedding
    assert np.allclose(guidance_scale_embedding(G["gse_w"], 256), G["gse_256"], atol=1e-6)
    assert np.allclose(guidance_scale_embedding(G["gse_w"], 255), G["gse_255"], atol=1e-6)
    lat, extra = draw_noise(42, 64, 64, 1)
    assert np.array_equal(lat.numpy(), G["latents_seed42_512x512"])
    assert np.array_equal(extra[0].numpy(), G["noise1_seed42_512x512"])


def test_schedule_matches_oracle_and_config_override(tmp_path):
    from oracle.scheduler import LCMSchedulerOracle
    from sdlcm_amd.scheduler import LCMSchedule
    s, o = LCMSchedule(), LCMSchedulerOracle()
    for n in (1, 2, 4, 8, 50):
        ts = s.timesteps(n)
        assert list(ts) == list(o.set_timesteps(n))
        for i in range(n):
            c, last = s.step_coefficients(ts, i)
            oc = o.coefficients(i)
            assert np.allclose(c, oc[:6], rtol=1e-7) and last == oc[6]
    with pytest.raises(ValueError):
        s.timesteps(51)
    p = tmp_path / "scheduler_config.json"
    p.write_text('{"beta_start": 0.001, "beta_end": 0.02, "original_inference_steps": 100, "_class_name": "X"}')
    s2 = LCMSchedule.from_config_file(str(p))
    assert s2.original_inference_steps == 100 and list(s2.timesteps(2)) == [999, 499]
    assert LCMSchedule.from_config_file(str(tmp_path / "missing.json")).original_inference_steps == 50


def test_weight_inventory_matches_known_parameter_counts():
    from sdlcm_amd import weights
    assert weights.count_params(weights.unet_param_spec()) == 859_602_884      # 859.52 M + cond_proj 81,920
    assert weights.count_params(weights.unet_param_spec(dict(time_cond_proj_dim=None))) == 859_520_964
    assert weights.count_params(weights.vae_param_spec()) == 49_490_199
    names = [n for n, _, _ in weights.unet_param_spec()]
    assert len(names) == len(set(names)) and "up_blocks.3.attentions.2.proj_out.weight" in names


def test_packing_layouts():
    from sdlcm_amd.packing import pack_conv3x3, pack_geglu
    w = torch.arange(2 * 3 * 9, dtype=torch.float32).reshape(2, 3, 3, 3)
    p = pack_conv3x3(w)
    assert p.shape == (2, 27)
    for tap in range(9):
        assert torch.equal(p[1, tap * 3:(tap + 1) * 3], w[1, :, tap // 3, tap % 3])
    wg = torch.arange(64 * 2, dtype=torch.float32).reshape(64, 2)
    bg = torch.arange(64, dtype=torch.float32)
    wp, bp = pack_geglu(wg, bg)
    # packed row 32q+r (r<16) = value row 16q+r ; 32q+16+r = gate row 32+16q+r
    for q in range(2):
        assert torch.equal(wp[32 * q:32 * q + 16], wg[16 * q:16 * q + 16])
        assert torch.equal(wp[32 * q + 16:32 * q + 32], wg[32 + 16 * q:32 + 16 * q + 16])
        assert torch.equal(bp[32 * q + 16:32 * q + 32], bg[32 + 16 * q:32 + 16 * q + 16])
    # GEGLU output / ff.net.2 columns in operand order: a permutation; channel 16P + 4fq + j at column 32(P>>1) + 8fq + 4(P&1) + j
    # (csrc/igemm_common.h geglu_store_col); 8 consecutive columns = the two quads one MFMA lane holds
    from sdlcm_amd.packing import geglu_col_order, pack_ff2_cols
    order = geglu_col_order(1280)
    assert sorted(order.tolist()) == list(range(1280))
    for ch in (0, 5, 17, 36, 1279):
        P, fq, j = ch // 16, (ch % 16) // 4, ch % 4
        assert order[32 * (P >> 1) + 8 * fq + 4 * (P & 1) + j] == ch
    assert order[:8].tolist() == [0, 1, 2, 3, 16, 17, 18, 19]
    w2 = torch.arange(3 * 64, dtype=torch.float32).reshape(3, 64)
    assert torch.equal(pack_ff2_cols(w2)[:, :8], w2[:, [0, 1, 2, 3, 16, 17, 18, 19]])


def test_worker_interface_mirror_and_errors(monkeypatch):
    from sdlcm_amd.backends import base, hip_worker, worker_factory
    from sdlcm_amd.lib import LcmHipError
    assert set(base.PipelineWorker.__annotations__) == {"worker_id"}
    assert hasattr(base.PipelineWorker, "run_job") and hasattr(base.PipelineWorker, "run_job_with_latents")
    assert hip_worker.parse_size("512X768") == (512, 768)
    with pytest.raises(RuntimeError, match="Invalid size 'bad', expected 'WIDTHxHEIGHT'"):
        hip_worker.parse_size("bad")
    monkeypatch.delenv("MODEL_ROOT", raising=False)
    monkeypatch.delenv("MODEL", raising=False)
    with pytest.raises(RuntimeError, match="MODEL_ROOT"):
        worker_factory.create_hip_worker(worker_id=0)
    monkeypatch.setenv("MODEL_ROOT", "/nonexistent")
    with pytest.raises(RuntimeError, match="MODEL "):
        worker_factory.create_hip_worker(worker_id=0)
    monkeypatch.setenv("MODEL", "nope")
    with pytest.raises(RuntimeError, match="not found"):
        worker_factory.create_hip_worker(worker_id=0)
    if not torch.cuda.is_available():
        monkeypatch.setenv("MODEL", "synthetic")
        with pytest.raises(LcmHipError, match="no CPU fallback"):       # product path fails loudly, never falls back
            worker_factory.create_hip_worker(worker_id=0)
    # CUDA_DTYPE: the reference honours fp16 / bf16 / fp32 and rejects anything else (backends/cuda_worker.py:55-61); this backend
    # has one arithmetic, so bf16 / fp32 are refused -- not silently run in fp16 -- unless LCM_HIP_DTYPE=fp16 says so
    monkeypatch.setenv("MODEL", "synthetic")
    monkeypatch.setenv("CUDA_DTYPE", "fp8")
    with pytest.raises(RuntimeError, match="Unknown CUDA_DTYPE=fp8"):
        worker_factory.create_hip_worker(worker_id=0)
    for dt in ("fp32", "bf16"):
        monkeypatch.setenv("CUDA_DTYPE", dt)
        monkeypatch.delenv("LCM_HIP_DTYPE", raising=False)
        with pytest.raises(RuntimeError, match=f"CUDA_DTYPE={dt} is not available with BACKEND=hip"):
            worker_factory.create_hip_worker(worker_id=0)
        if not torch.cuda.is_available():
            monkeypatch.setenv("LCM_HIP_DTYPE", "fp16")                  # explicit override: construction proceeds (to the GPU check)
            with pytest.raises(LcmHipError, match="no CPU fallback"):
                worker_factory.create_hip_worker(worker_id=0)


def test_png_encoding_is_deterministic_and_lossless():
    import io
    from PIL import Image
    from sdlcm_amd.backends.hip_worker import encode_png
    rgb = np.random.RandomState(0).randint(0, 256, size=(64, 48, 3), dtype=np.uint8)
    a, b = encode_png(rgb), encode_png(rgb)
    assert a == b and a[:8] == b"\x89PNG\r\n\x1a\n"
    im = Image.open(io.BytesIO(a))
    im.verify()                                    # chunk CRCs / structure
    assert np.array_equal(np.asarray(Image.open(io.BytesIO(a))), rgb)
    yy, xx = np.mgrid[0:40, 0:56]
    smooth = np.stack([xx * 4, yy * 6, (xx + yy) * 2], -1).astype(np.uint8)      # wrap-around rows exercise the Up filter
    assert np.array_equal(np.asarray(Image.open(io.BytesIO(encode_png(smooth)))), smooth)
    os.environ["LCM_PNG_ENCODER"] = "pil"
    try:
        assert np.array_equal(np.asarray(Image.open(io.BytesIO(encode_png(rgb)))), rgb)
    finally:
        del os.environ["LCM_PNG_ENCODER"]


def test_png_parallel_deflate_is_one_valid_zlib_stream():
    """Stripes deflated on several threads, concatenated behind one zlib header: the IDAT payload must inflate (header,
    every segment boundary, the Adler-32 trailer) to exactly the filtered scanlines, for any stripe count and for heights
    that do not divide evenly; the bytes are a function of the configuration, not of thread timing."""
    import io, struct, zlib
    from PIL import Image
    from sdlcm_amd.backends.hip_worker import encode_png
    rs = np.random.RandomState(1)
    yy, xx = np.mgrid[0:360, 0:640]
    img = (np.stack([xx // 3, yy // 2, (xx + yy) // 4], -1) + rs.randint(0, 6, size=(360, 640, 3))).astype(np.uint8)
    old = os.environ.get("LCM_PNG_THREADS")
    try:
        files = {}
        for thr in ("1", "2", "4", "5"):
            os.environ["LCM_PNG_THREADS"] = thr
            a = encode_png(img)
            assert a == encode_png(img)
            Image.open(io.BytesIO(a)).verify()
            assert np.array_equal(np.asarray(Image.open(io.BytesIO(a))), img)
            n = struct.unpack(">I", a[33:37])[0]
            assert a[37:41] == b"IDAT"
            raw = zlib.decompress(a[41:41 + n])                      # strict: checks the Adler-32 trailer
            assert len(raw) == 360 * (1 + 640 * 3) and raw[0] == 2
            files[thr] = a
        assert files["1"] != files["4"]                              # really striped
        os.environ["LCM_PNG_THREADS"] = "4"
        small = rs.randint(0, 256, size=(64, 64, 3), dtype=np.uint8)  # under 128 scanlines: a single stripe
        assert np.array_equal(np.asarray(Image.open(io.BytesIO(encode_png(small)))), small)
    finally:
        if old is None:
            os.environ.pop("LCM_PNG_THREADS", None)
        else:
            os.environ["LCM_PNG_THREADS"] = old


def test_png_writer_of_the_library_handles_every_block_form():
    """csrc/png.cpp through the C ABI: noise (stored blocks: Huffman would not pay), flat colour (distance-1 runs), vertical
    gradients (distance-3 repeats), sparse symbols (code lengths limited to 15 / 7 bits), 1-pixel and odd-sized images, a
    pitch wider than the row; every file must inflate strictly (Adler-32) and decode to the input; a short output buffer is
    refused before anything is written; the round-3 zlib writer stays selectable."""
    import ctypes, io, struct, zlib
    from PIL import Image
    from sdlcm_amd import lib
    from sdlcm_amd.backends.hip_worker import encode_png
    L = lib.load()
    rs = np.random.RandomState(2)
    yy, xx = np.mgrid[0:200, 0:150]
    skew = np.zeros((200, 150, 3), np.uint8)
    skew[rs.rand(200, 150, 3) < 0.001] = rs.randint(1, 256)          # one dominant symbol, a few rare ones
    ramp = (rs.geometric(0.5, size=(200, 150, 3)).clip(0, 40) * 6).astype(np.uint8)      # frequencies falling by powers of two
    cases = {"noise": rs.randint(0, 256, size=(200, 150, 3), dtype=np.uint8),
             "flat": np.full((200, 150, 3), 77, np.uint8),
             "vgrad": np.stack([yy, yy * 2, yy * 3], -1).astype(np.uint8),
             "hgrad": np.stack([xx, xx * 2, xx // 2], -1).astype(np.uint8),
             "skew": skew, "ramp": ramp,
             "1x1": rs.randint(0, 256, size=(1, 1, 3), dtype=np.uint8),
             "3x2": rs.randint(0, 256, size=(3, 2, 3), dtype=np.uint8)}
    for name, img in cases.items():
        for stripes in (1, 3):
            h, w, _ = img.shape
            cap = L.lcm_png_bound(w, h, stripes)
            out = np.empty(cap, np.uint8)
            n = ctypes.c_longlong(0)
            assert L.lcm_png_encode_rgb8(img.ctypes.data, w, h, w * 3, stripes, out.ctypes.data, cap, ctypes.byref(n)) == 0, name
            a = out[:n.value].tobytes()
            Image.open(io.BytesIO(a)).verify()
            assert np.array_equal(np.asarray(Image.open(io.BytesIO(a))), img), name
            ln = struct.unpack(">I", a[33:37])[0]
            raw = zlib.decompress(a[41:41 + ln])
            assert len(raw) == h * (1 + w * 3), name
            assert n.value <= cap
    assert len(encode_png(cases["flat"])) < 1000 and len(encode_png(cases["vgrad"])) < 1500     # the matches are found
    # a pitch wider than the row: a view into a larger image
    big = rs.randint(0, 256, size=(96, 128, 3), dtype=np.uint8)
    view = big[:, 16:80]
    out = np.empty(L.lcm_png_bound(64, 96, 1), np.uint8)
    n = ctypes.c_longlong(0)
    assert L.lcm_png_encode_rgb8(view.ctypes.data, 64, 96, 128 * 3, 1, out.ctypes.data, out.size, ctypes.byref(n)) == 0
    assert np.array_equal(np.asarray(Image.open(io.BytesIO(out[:n.value].tobytes()))), view)
    assert L.lcm_png_encode_rgb8(view.ctypes.data, 64, 96, 128 * 3, 1, out.ctypes.data, 1000, ctypes.byref(n)) != 0
    assert b"output buffer" in L.lcm_last_error()
    assert L.lcm_png_encode_rgb8(view.ctypes.data, 64, 96, 100, 1, out.ctypes.data, out.size, ctypes.byref(n)) != 0       # pitch < row
    os.environ["LCM_PNG_ENCODER"] = "zlib"
    try:
        assert np.array_equal(np.asarray(Image.open(io.BytesIO(encode_png(big)))), big)
    finally:
        del os.environ["LCM_PNG_ENCODER"]


def test_bpe_tokenizer_id_contract(tmp_path):
    """_BpeTokenizer against a vocabulary the test writes itself: BOS + BPE ids + EOS, padding to 77 with the directory's
    own pad id (SD1.5 / first SDXL tokenizer: EOS; second SDXL tokenizer: id 0), truncation to 77 keeping EOS last, int32 --
    the id contract of backends/rknnlcm.py:305-324."""
    import torch
    import tinytok
    from sdlcm_amd.prompt import _BpeTokenizer
    v = tinytok.vocab()
    bos, eos = v["<|startoftext|>"], v["<|endoftext|>"]
    t1 = _BpeTokenizer(tinytok.write(str(tmp_path / "tokenizer")))
    ids = t1(["A cat, a dog", "cat " * 200, ""])
    assert ids.dtype == torch.int32 and tuple(ids.shape) == (3, 77)
    assert ids[0, :7].tolist() == [bos, v["a</w>"], v["cat</w>"], v[",</w>"], v["a</w>"], v["dog</w>"], eos]       # lower-cased, merged
    assert (ids[0, 7:] == eos).all()
    assert ids[1, 0] == bos and ids[1, 76] == eos and (ids[1, 1:76] == v["cat</w>"]).all()                       # truncated to 77
    assert ids[2, 0] == bos and (ids[2, 1:] == eos).all()
    t2 = _BpeTokenizer(tinytok.write(str(tmp_path / "tokenizer_2"), pad="!"))
    ids2 = t2(["a cat"])
    assert ids2.dtype == torch.int32 and ids2[0, :4].tolist() == [bos, v["a</w>"], v["cat</w>"], eos]
    assert v["!"] == 0 and (ids2[0, 4:] == 0).all()                                                                  # SDXL tokenizer_2 pads with id 0


def test_real_text_encoder_weights_need_a_vocabulary(tmp_path, monkeypatch):
    """make_tokenizer: checkpoint directory first, then LCM_TOKENIZER_DIR; real weights without a vocabulary raise (never the
    hashed stand-in); synthetic weights fall back to HashTokenizer."""
    import tinytok
    from sdlcm_amd.clip import HashTokenizer
    from sdlcm_amd.lib import LcmHipError
    from sdlcm_amd.prompt import _BpeTokenizer, make_tokenizer
    monkeypatch.delenv("LCM_TOKENIZER_DIR", raising=False)
    assert isinstance(make_tokenizer(None, "tokenizer", 49408, False), HashTokenizer)
    with pytest.raises(LcmHipError, match="LCM_TOKENIZER_DIR"):
        make_tokenizer(None, "tokenizer", 49408, True)
    root = tmp_path / "ckpt"
    (root / "tokenizer").mkdir(parents=True)                      # an empty directory is not a vocabulary
    with pytest.raises(LcmHipError, match="no CLIP vocabulary"):
        make_tokenizer(str(root), "tokenizer", 49408, True)
    tinytok.write(str(root / "tokenizer"))
    assert isinstance(make_tokenizer(str(root), "tokenizer", 49408, True), _BpeTokenizer)
    env = tmp_path / "vocab"
    tinytok.write(str(env / "tokenizer"))
    tinytok.write(str(env / "tokenizer_2"), pad="!")
    monkeypatch.setenv("LCM_TOKENIZER_DIR", str(env))
    assert isinstance(make_tokenizer(None, "tokenizer", 49408, True), _BpeTokenizer)
    assert int(make_tokenizer(None, "tokenizer_2", 49408, True)(["x"])[0, -1]) == 0
    monkeypatch.setenv("LCM_TOKENIZER_DIR", str(env / "tokenizer"))           # the vocabulary files themselves: first tokenizer only
    assert isinstance(make_tokenizer(None, "tokenizer", 49408, True), _BpeTokenizer)
    with pytest.raises(LcmHipError):
        make_tokenizer(None, "tokenizer_2", 49408, True)


def test_launch_hooks_are_thread_local():
    import threading
    from sdlcm_amd import ops
    seen = []
    with ops.recording() as recs, ops.profiling() as prof:
        assert ops._record_list() is recs and ops._profile_list() is prof
        t = threading.Thread(target=lambda: seen.append((ops._record_list(), ops._profile_list())))
        t.start(); t.join()
        with ops.recording() as inner:             # nesting restores the outer list
            assert ops._record_list() is inner
        assert ops._record_list() is recs
    assert seen == [(None, None)] and ops._record_list() is None and ops._profile_list() is None


def test_hash_tokenizer_layout():
    from sdlcm_amd.clip import HashTokenizer, clip_param_spec
    from sdlcm_amd import weights
    t = HashTokenizer()
    a, b, c = t(["a cat"]), t(["a cat"]), t(["a dog " * 60])
    assert a.shape == (1, 77) and a.dtype == torch.int32 and torch.equal(a, b)
    assert a[0, 0] == 49406 and a[0, 3] == 49407 and (a[0, 3:] == 49407).all()          # BOS w1 w2 EOS pad...
    assert c[0, 0] == 49406 and c[0, 76] == 49407 and c.shape == (1, 77)                 # truncated to 77
    assert weights.count_params(clip_param_spec()) == 123_060_480                          # CLIP-L text model


def test_lora_key_parsing_kohya_and_peft():
    from sdlcm_amd.config import unet_config
    from sdlcm_amd.lora import parse_lora
    cfg = unet_config(dict(block_out_channels=(64, 128, 128, 128), attention_head_dim=8, cross_attention_dim=96))
    r = 4
    raw = {
        # kohya naming: dots flattened to underscores, separate alpha
        "lora_unet_down_blocks_0_attentions_1_transformer_blocks_0_attn1_to_q.lora_down.weight": torch.randn(r, 64),
        "lora_unet_down_blocks_0_attentions_1_transformer_blocks_0_attn1_to_q.lora_up.weight": torch.randn(64, r),
        "lora_unet_down_blocks_0_attentions_1_transformer_blocks_0_attn1_to_q.alpha": torch.tensor(2.0),
        "lora_unet_mid_block_attentions_0_proj_in.lora_down.weight": torch.randn(r, 128, 1, 1),
        "lora_unet_mid_block_attentions_0_proj_in.lora_up.weight": torch.randn(128, r, 1, 1),
        # peft / diffusers naming
        "unet.up_blocks.1.attentions.2.transformer_blocks.0.attn2.to_k.lora_A.weight": torch.randn(r, 96),
        "unet.up_blocks.1.attentions.2.transformer_blocks.0.attn2.to_k.lora_B.weight": torch.randn(128, r),
        # text encoder entries are not the UNet parser's (lora.ClipLora takes them); conv_in IS a target, but this
        # entry has no lora_up partner, so it is matched and then dropped as incomplete
        "lora_te_text_model_encoder_layers_0_mlp_fc1.lora_down.weight": torch.randn(r, 768),
        "lora_unet_conv_in.lora_down.weight": torch.randn(r, 4, 3, 3),
        "lora_unet_conv_out.lora_down.weight": torch.randn(r, 64, 3, 3),
        "lora_unet_conv_out.lora_up.weight": torch.randn(4, r, 1, 1),
        # LoCon on a resnet conv: 3x3 extent on lora_down, 1x1 lora_up
        "lora_unet_down_blocks_0_resnets_0_conv1.lora_down.weight": torch.randn(r, 64, 3, 3),
        "lora_unet_down_blocks_0_resnets_0_conv1.lora_up.weight": torch.randn(64, r, 1, 1),
    }
    parsed, skipped = parse_lora(raw, cfg)
    assert set(parsed) == {"down_blocks.0.attentions.1.transformer_blocks.0.attn1.to_q", "mid_block.attentions.0.proj_in",
                           "up_blocks.1.attentions.2.transformer_blocks.0.attn2.to_k", "down_blocks.0.resnets.0.conv1", "conv_out"}
    assert parsed["conv_out"][0].shape == (r, 64, 3, 3) and parsed["conv_out"][1].shape == (4, r)
    assert parsed["down_blocks.0.resnets.0.conv1"][0].shape == (r, 64, 3, 3) and parsed["down_blocks.0.resnets.0.conv1"][1].shape == (64, r)
    d, u, a = parsed["down_blocks.0.attentions.1.transformer_blocks.0.attn1.to_q"]
    assert d.shape == (r, 64) and u.shape == (64, r) and a == 2.0
    assert parsed["mid_block.attentions.0.proj_in"][0].shape == (r, 128) and parsed["mid_block.attentions.0.proj_in"][2] == float(r)
    assert len(skipped) == 1


def test_style_registry_mirror():
    from sdlcm_amd.backends.styles import STYLE_REGISTRY, StyleDef
    sd = STYLE_REGISTRY["papercut"]
    assert sd.adapter_name == "style_papercut" and sd.weight_for(1) == 0.80 and sd.weight_for(99) == 1.15 and sd.weight_for(0) == 0.80
    os.environ["LCM_STYLE_PAPERCUT_PATH"] = "/tmp/x.safetensors"
    try:
        assert sd.path() == "/tmp/x.safetensors"
    finally:
        del os.environ["LCM_STYLE_PAPERCUT_PATH"]


# ---- queue-level micro-batching: host logic only ------------------------------------
def test_microbatcher_coalesces_by_key_and_keeps_order():
    import threading, time
    from sdlcm_amd.backends.batching import MicroBatcher
    gate, calls = threading.Event(), []

    def run(key, items):
        if key == "block":
            gate.wait(5)
        calls.append((key, list(items)))
        return [f"{key}:{it}" for it in items]

    mb = MicroBatcher(run, max_batch=8)
    first = mb.submit("block", 0)                 # occupies the dispatcher while the queue fills
    time.sleep(0.05)
    futs = [mb.submit("a", i) for i in range(5)] + [mb.submit("b", 9)] + [mb.submit("a", 5)]
    gate.set()
    assert first.result(5) == "block:0"
    assert [f.result(5) for f in futs] == ["a:0", "a:1", "a:2", "a:3", "a:4", "b:9", "a:5"]
    # 6 "a" jobs were waiting: plan batch sizes are 1/2/4/8 -> 4 first, then "b" is NOT starved behind the 2 left
    assert calls[1] == ("a", [0, 1, 2, 3])
    assert ("b", [9]) in calls and ("a", [4, 5]) in calls
    assert mb.batches == [1, 4, 1, 2] or mb.batches == [1, 4, 2, 1]
    mb.close()
    import pytest
    with pytest.raises(RuntimeError):
        mb.submit("a", 1)


def test_microbatcher_errors_reach_every_waiter_and_window():
    import time
    from sdlcm_amd.backends.batching import MicroBatcher

    def run(key, items):
        if key == "bad":
            raise ValueError("boom")
        return items

    mb = MicroBatcher(run, max_batch=4, window_ms=200)
    t0 = time.monotonic()
    fa, fb = mb.submit("k", 1), mb.submit("k", 2)          # arrive within the window -> one pass of 2
    assert fa.result(5) == 1 and fb.result(5) == 2
    assert mb.batches == [2] and time.monotonic() - t0 >= 0.15
    bad = [mb.submit("bad", i) for i in range(2)]
    import pytest
    for f in bad:
        with pytest.raises(ValueError):
            f.result(5)
    assert mb.submit("k", 3).result(5) == 3                 # the dispatcher survives a failed pass
    mb.close()


def test_run_job_reproduces_the_reference_recordings():
    """HipLcmWorker.run_job against what DiffusersCudaWorker.run_job did with a recording fake ``pipe``
    (tests/golden/worker_contract.json, recorded from the reference by tests/golden/make_contract_golden.py): the values the
    sampler receives (prompt, width, height, steps, guidance -- with the reference's int() / float() coercions), the seed policy
    (request seed returned as is; none: a draw below 10^8), the style in effect DURING the pass (the adapter and weight the
    reference's _apply_style handed to the pipeline, after its level clamp; off for level 0 / unknown styles), no style state
    bleeding into the next job, and the error text of malformed sizes.  The engine is a recording stand-in: no GPU involved."""
    import json
    import types
    import numpy as np
    from sdlcm_amd.backends import hip_worker
    from sdlcm_amd.backends.styles import STYLE_REGISTRY
    from sdlcm_amd.pipeline import check_size
    from sdlcm_amd.lib import LcmHipError
    doc = json.load(open(os.path.join(ROOT, "tests", "golden", "worker_contract.json")))
    passes = []

    class FakeEngine:
        batcher, batch_sizes = None, (1,)
        pipe = types.SimpleNamespace(sched=types.SimpleNamespace(init_noise_sigma=1.0))
        styles = {sid: object() for sid in STYLE_REGISTRY}                 # every registered style "loaded", as in the recording
        _want_style = hip_worker._Engine._want_style

        def run_batch(self, key, items):
            passes.append((key, items))
            return [(np.zeros((8, 8, 3), np.uint8), np.zeros((1, 4, 8, 8), np.float16)) for _ in items]

    w = object.__new__(hip_worker.HipLcmWorker)
    w.worker_id, w._engine = 0, FakeEngine()
    adapter_to_style = {sd.adapter_name: sid for sid, sd in STYLE_REGISTRY.items()}
    for rec in doc["run_job"]:
        r = dict(rec["request"])
        sl = r.pop("style_lora", None)
        req = types.SimpleNamespace(**r)
        if sl is not None:
            req.style_lora = types.SimpleNamespace(style=sl[0], level=sl[1])
        del passes[:]
        if "error" in rec:
            with pytest.raises(RuntimeError) as ei:
                w.run_job(types.SimpleNamespace(req=req))
            assert str(ei.value) == rec["error"] and not passes
            continue
        png, seed = w.run_job(types.SimpleNamespace(req=req))
        assert png[:8] == b"\x89PNG\r\n\x1a\n" and rec["png_magic_ok"]
        ref_pipe = [e for e in rec["events"] if e[0] == "pipe"][0][1]
        (key, items), = passes
        width, height, steps, guidance, style_id, level = key
        assert (width, height, steps, guidance) == (ref_pipe["width"], ref_pipe["height"], ref_pipe["num_inference_steps"], ref_pipe["guidance_scale"])
        assert items[0][0].prompt == ref_pipe["prompt"]
        if rec["seed_is_request_seed"]:
            assert seed == rec["returned_seed"] == ref_pipe["generator_initial_seed"] and items[0][1] == seed
        else:
            assert 0 <= seed < doc["seed_policy_without_seed"]["upper_bound_exclusive"] and items[0][1] == seed
        # the style in effect during the pass: what the reference's _apply_style handed to the pipeline before the call
        before = rec["events"][0]
        want = w._engine._want_style(style_id, level)
        if before[0] == "set_adapters":
            assert want is not None and [want[0]] == [adapter_to_style[a] for a in before[1]] and [want[1]] == before[2]
        else:
            assert before == ["disable_lora"] and want is None
        assert rec["events"][-1] == ["disable_lora"]         # reference: reset after every job; here the NEXT job's key carries its own style
    # no bleed: a styled job followed by a plain one -> the plain job's pass asks for no style
    del passes[:]
    w.run_job(types.SimpleNamespace(req=types.SimpleNamespace(prompt="s", size="64x64", num_inference_steps=1, guidance_scale=1.0, seed=1,
                                                                style_lora=types.SimpleNamespace(style="papercut", level=2))))
    w.run_job(types.SimpleNamespace(req=types.SimpleNamespace(prompt="p", size="64x64", num_inference_steps=1, guidance_scale=1.0, seed=2)))
    assert w._engine._want_style(*passes[0][0][4:]) is not None and w._engine._want_style(*passes[1][0][4:]) is None
    # the pipeline's size rule (check_inputs): accept / reject and the message text
    n = 0
    for rec in doc["check_inputs"]:
        a = rec["args"]
        if not (a.get("prompt") == "a" and a.get("callback_steps") == 1 and len(a) == 4):
            continue
        n += 1
        if rec["outcome"] == "ok":
            check_size(a["width"], a["height"])
        else:
            with pytest.raises(LcmHipError) as ei:
                check_size(a["width"], a["height"])
            assert str(ei.value) == rec["error"]
    assert n >= 4


def test_workers_map_to_devices_like_the_reference_maps_npu_cores():
    """LCM_DEVICES=all: worker i -> cuda:(i mod N), the reference's rule for its accelerator cores
    (server/lcm_sr_server.py:140-152); a list picks from the list; unset keeps the single-device behaviour."""
    from sdlcm_amd.backends.worker_factory import pick_device
    assert [pick_device(i, 8, {"LCM_DEVICES": "all"}) for i in range(10)] == [f"cuda:{i % 8}" for i in range(10)]
    assert [pick_device(i, 8, {"LCM_DEVICES": "1,3,6"}) for i in range(4)] == ["cuda:1", "cuda:3", "cuda:6", "cuda:1"]
    assert pick_device(5, 8, {}) == "cuda:0" and pick_device(5, 8, {"CUDA_DEVICE": "cuda:2"}) == "cuda:2"
    assert pick_device(5, 8, {"HIP_DEVICE": "cuda:4", "CUDA_DEVICE": "cuda:2"}) == "cuda:4"
    assert pick_device(3, 1, {"LCM_DEVICES": "all"}) == "cuda:0"
    for bad in ("9", "0,x", "-1"):
        with pytest.raises(RuntimeError, match="LCM_DEVICES"):
            pick_device(0, 8, {"LCM_DEVICES": bad})
    with pytest.raises(RuntimeError, match="no GPU"):
        pick_device(0, 0, {"LCM_DEVICES": "all"})


def test_second_lane_serves_when_lane0_is_stuck():
    """Lanes above 0 leave a long queue to lane 0's next (larger) batch -- but only while lane 0 actually comes back for it: when
    lane 0 sits inside one call for long (first-use tune / graph capture, style wait) the second lane takes the work."""
    import threading, time
    from sdlcm_amd.backends.batching import MicroBatcher
    gate, served = threading.Event(), []

    def run(key, items, lane):
        if key == "stuck":
            gate.wait(10)
        served.append((lane, key, len(items), time.monotonic()))
        return list(items)

    mb = MicroBatcher(run, max_batch=8, lanes=2)
    mb.lane0_stall_s = 0.1
    t0 = time.monotonic()
    first = mb.submit("stuck", 0)
    time.sleep(0.05)
    assert mb._lane0_busy                        # an idle lane 0 always takes the head
    futs = [mb.submit("a", i) for i in range(6)]  # more than lane_max_waiting: lane 1 defers at first ...
    assert [f.result(5) for f in futs] == list(range(6))            # ... and serves once lane 0 has been away > lane0_stall_s
    assert all(lane == 1 for lane, key, n, t in served if key == "a") and time.monotonic() - t0 < 3.0
    gate.set()
    assert first.result(5) == 0
    mb.close()


def test_second_lane_takes_a_full_batch_waiting_behind_lane0():
    """With more than a couple of jobs waiting the second lane normally leaves them to lane 0's next batched pass -- unless a
    FULL batch is already waiting while lane 0 is inside a pass: then it runs that batch concurrently."""
    import threading, time
    from sdlcm_amd.backends.batching import MicroBatcher
    gate, served = threading.Event(), []

    def run(key, items, lane):
        if lane == 0:
            gate.wait(10)
        served.append((lane, len(items)))
        return list(items)

    mb = MicroBatcher(run, max_batch=8, lanes=2)
    mb.lane0_stall_s = 5.0                           # the stall rule must not be what lets lane 1 in
    first = mb.submit("k", -1)
    time.sleep(0.05)
    assert mb._lane0_busy
    few = [mb.submit("k", i) for i in range(5)]      # 5 waiting: not a full batch -> lane 1 defers
    time.sleep(0.3)
    assert not served
    more = [mb.submit("k", 5 + i) for i in range(3)] # 8 waiting: lane 1 takes them as one batch of 8 while lane 0 is still busy
    assert [f.result(5) for f in few + more] == list(range(8))
    assert served == [(1, 8)] and not first.done()
    gate.set()
    assert first.result(5) == -1
    mb.close()


def test_bench_keeps_stdout_for_one_json_line():
    """bench.py's contract is ONE JSON line on stdout; RCCL prints its version banner to stdout at communicator setup.  After
    claim_stdout() everything written to fd 1 -- Python prints and C-level writes -- lands on stderr; emit_line() alone reaches
    the real stdout."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, sys; sys.path.insert(0, %r); import bench\n"
            "fd = bench.claim_stdout()\n"
            "print('python chatter'); os.write(1, b'C-level banner\\n'); os.system('echo child chatter')\n"
            "bench.emit_line(fd, {'metric': 'm', 'value': 1.5})\n" % root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout.count("\n") == 1 and json.loads(r.stdout) == {"metric": "m", "value": 1.5}
    for chatter in ("python chatter", "C-level banner", "child chatter"):
        assert chatter in r.stderr


def test_lone_caller_stays_on_lane0():
    """One caller at a time never reaches the other lanes (each would allocate its own workspace, buffers and graphs on first
    use): lane k > 0 only serves while lane 0 is inside a pass.  Two concurrent callers do use both."""
    import threading, time
    from sdlcm_amd.backends.batching import MicroBatcher
    served = []

    def run(key, items, lane):
        time.sleep(0.02)
        served.append(lane)
        return list(items)

    mb = MicroBatcher(run, max_batch=8, lanes=2)
    for i in range(25):
        assert mb.submit("k", i).result(5) == i
    assert set(served) == {0}
    served.clear()
    out = []
    ts = [threading.Thread(target=lambda j=j: out.extend(mb.submit("k", 100 * j + i).result(5) for i in range(10))) for j in range(2)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert sorted(out) == sorted(100 * j + i for j in range(2) for i in range(10))
    assert set(served) == {0, 1}
    mb.close()


def test_upsample_phase_packing_matches_interpolate_then_conv():
    """packing.pack_conv3x3_up2: nearest-2x upsample -> conv3x3 == four 2x2 phase convolutions on the low-resolution
    input with the coinciding taps summed (SURVEY A.5 Upsample2D); checked in fp32 against the torch op order."""
    import torch
    import torch.nn.functional as F
    from sdlcm_amd.packing import pack_conv3x3_up2
    g = torch.Generator().manual_seed(0)
    Cin, Cout, H, W = 8, 6, 5, 7
    w = torch.randn(Cout, Cin, 3, 3, generator=g)
    x = torch.randn(2, Cin, H, W, generator=g)
    ref = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, padding=1)
    wp = pack_conv3x3_up2(w).reshape(4, Cout, 2, 2, Cin)
    xp = F.pad(x, (1, 1, 1, 1))
    out = torch.zeros_like(ref)
    for py in range(2):
        for px in range(2):
            acc = torch.zeros(2, Cout, H, W)
            for dy in range(2):
                for dx in range(2):
                    acc += torch.einsum("bchw,oc->bohw", xp[:, :, py + dy:py + dy + H, px + dx:px + dx + W], wp[py * 2 + px, :, dy, dx, :])
            out[:, :, py::2, px::2] = acc
    assert (out - ref).abs().max() < 1e-4


def test_plan_table_sources_and_gn_fusion_rule(tmp_path, monkeypatch):
    """Launch plans: the shipped table is read unless LCM_TUNED_PLANS=0, a user LCM_TUNE_CACHE file overlays it; the
    GroupNorm-into-conv fusion rule only fires for tensors beyond the Infinity Cache and convs with few n-tiles."""
    import json
    from sdlcm_amd import autotune, model
    shipped = autotune._load_cache()
    assert len(shipped) > 100 and all(len(k) == 5 and v[0] in (64, 128) and v[1] in (64, 128, 160) for k, v in shipped.items())
    key = next(iter(shipped))
    user = tmp_path / "plans.json"
    user.write_text(json.dumps({",".join(str(x) for x in key): [64, 64, 1, -1, 0.5], "0,7,64,64,1": [64, 64, 1, -1, 0.1]}))
    monkeypatch.setenv("LCM_TUNE_CACHE", str(user))
    merged = autotune._load_cache()
    assert tuple(merged[key][:4]) == (64, 64, 1, -1) and (0, 7, 64, 64, 1) in merged and len(merged) == len(shipped) + 1
    monkeypatch.setenv("LCM_TUNED_PLANS", "0")
    assert set(autotune._load_cache()) == {key, (0, 7, 64, 64, 1)}
    assert model._fuse_gn_into_conv(8 * 512 * 512, 128, 128)            # VAE 512^2 level at batch 8: 537 MB, one n-tile
    assert model._fuse_gn_into_conv(512 * 512, 128, 128)                # ... and at batch 1 (67 MB)
    assert not model._fuse_gn_into_conv(4096, 320, 320)                 # batch-1 UNet level: 2.6 MB, served from cache
    assert not model._fuse_gn_into_conv(8 * 128 * 128, 512, 512)        # 4 n-tiles of 128: the transform would repeat 4x


def test_bench_traffic_lookup_reads_committed_pmc_table(tmp_path, capsys):
    """bench.py's roofline.traffic comes from the committed PMC table (separate --pmc passes): 2 x FETCH_SIZE + WRITE_SIZE
    in bytes for a profiled (kernel, batch); for anything else None WITH the reason (named in traffic_source and on
    stderr) -- never a guess, never a silent null."""
    import importlib.util, json, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from tools.check_profiles_fresh import CSRC, sources_sha256
    files = [os.path.join(CSRC, "common.h")]
    tabf = tmp_path / "traffic.json"
    doc = {"batch": {"1": {"void k<1>(P)": {"FETCH_SIZE": 100.0, "WRITE_SIZE": 50.0, "n_FETCH_SIZE": 7}}},
           "csrc_files": files, "csrc_sha256": sources_sha256(files)}
    tabf.write_text(json.dumps(doc))
    bench.TRAFFIC_FILE = str(tabf)
    t, src = bench._pmc_traffic("void k<1>(P)", 1)
    assert t == int((2 * 100.0 + 50.0) * 1024) and "x2" in src
    for args in (("no_such_kernel<1>", 1), ("void k<1>(P)", 3)):
        t, src = bench._pmc_traffic(*args)
        assert t is None and "absent" in src
    assert "WARNING" in capsys.readouterr().err
    bench.TRAFFIC_FILE = str(tmp_path / "missing.json")
    t, src = bench._pmc_traffic("void k<1>(P)", 1)
    assert t is None and "unreadable" in src
    # a table taken on other kernel sources reports nothing
    stale = tmp_path / "stale.json"
    stale.write_text(json.dumps(dict(doc, csrc_sha256="0" * 64)))
    bench.TRAFFIC_FILE = str(stale)
    t, src = bench._pmc_traffic("void k<1>(P)", 1)
    assert t is None and "other kernel sources" in src
    # counters: unique match required, stale entries nulled
    summ = tmp_path / "summary.json"
    ent = dict(what="w", source="s", sources=files, source_sha256=sources_sha256(files), avg_us=100.0, mfma_busy_frac=0.4,
               mfma_busy_frac_at_2p1ghz=0.3, effective_clock_ghz=1.9, hbm_gbs=1000.0)
    summ.write_text(json.dumps({"attn2_kernel<40, 8, 1>": ent, "attn2_kernel<40, 8, 2>": ent, "k<1, 2>": dict(ent, source_sha256="0" * 64)}))
    bench.PMC_SUMMARY_FILE = str(summ)
    assert bench._pmc_counters("attn2_kernel<40, 8, 2>")["mfma_busy_frac"] == 0.4
    amb = bench._pmc_counters("attn2_kernel<40, 8")
    assert amb["mfma_busy_frac"] is None and "matches 2 entries" in amb["pmc_source"]
    st = bench._pmc_counters("k<1, 2>")
    assert st["mfma_busy_frac"] is None and "stale" in st["pmc_source"]


def test_committed_counter_evidence_was_taken_on_the_kernels_in_the_tree():
    """PMC evidence went stale when a kernel changed after its counters were collected.  Every entry of this round's
    profiles/r04_pmc_summary.json and the traffic table record the sha256 of the kernel sources they were taken on
    (tools/pmc_summary.py, tools/pmc_traffic.py); editing a kernel without re-collecting (tools/pmc_round4.sh, tools/pmc_traffic.sh)
    fails here."""
    from tools.check_profiles_fresh import ROOT as R, stale_entries
    p = os.path.join(R, "profiles", "r04_pmc_summary.json")
    assert os.path.exists(p), "profiles/r04_pmc_summary.json missing"
    bad = stale_entries(p)
    assert not bad, f"stale PMC entries (re-run tools/pmc_round4.sh + tools/pmc_summary.py r04): {bad}"
    import subprocess, sys
    r = subprocess.run([sys.executable, os.path.join(R, "tools", "check_profiles_fresh.py"), "r04"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout


def test_stream_workspace_entries_are_owned_by_their_pointer():
    """workspaces are keyed by launch stream, and framework streams are recycled handles -- a second owner on one handle
    must be refused, and one owner's unregister must not erase another's entry.  Pure host bookkeeping: no GPU involved."""
    import ctypes as C
    from sdlcm_amd import lib
    L = lib.load()
    s1, a, b = C.c_void_p(0x1000), C.c_void_p(0xA000), C.c_void_p(0xB000)
    assert L.lcm_set_stream_workspace(s1, a, 1 << 20) == 0
    assert L.lcm_set_stream_workspace(s1, a, 2 << 20) == 0                 # the owner may re-register (grow)
    assert L.lcm_set_stream_workspace(s1, b, 1 << 20) != 0                 # another owner on the same handle: refused
    assert b"another owner" in L.lcm_last_error()
    assert L.lcm_set_stream_workspace(s1, b, 0) == 0                       # b forgetting "its" entry leaves a's alone ...
    assert L.lcm_set_stream_workspace(s1, b, 1 << 20) != 0                 # ... (still a's)
    assert L.lcm_set_stream_workspace(s1, a, 0) == 0                       # a forgets its own
    assert L.lcm_set_stream_workspace(s1, b, 1 << 20) == 0                 # now the handle is free
    assert L.lcm_set_stream_workspace(s1, None, 0) == 0                    # unconditional
    assert L.lcm_set_stream_workspace(s1, a, 1 << 20) == 0
    assert L.lcm_set_stream_workspace(s1, None, 0) == 0
