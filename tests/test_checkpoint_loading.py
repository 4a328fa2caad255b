"""Checkpoint path: a diffusers-layout directory (model_index.json, unet/, vae/,
scheduler/) round-trips through the loader with the graph-vs-checkpoint shape audit, and the worker factory reads
cross_attention_dim from it like backends/worker_factory.py:55-67.  Uses a narrow synthetic checkpoint written on the fly."""
import json
import os

import pytest
import torch


def _write_ckpt(root, ucfg, vcfg, vae_attn_names=("to_q", "to_k", "to_v", "to_out.0")):
    from safetensors.torch import save_file
    from sdlcm_amd import weights
    from sdlcm_amd.config import unet_config, vae_config
    os.makedirs(os.path.join(root, "unet"))
    os.makedirs(os.path.join(root, "vae"))
    os.makedirs(os.path.join(root, "scheduler"))
    usd = weights.synthetic_state_dict(weights.unet_param_spec(ucfg), 0)
    vsd = weights.synthetic_state_dict(weights.vae_param_spec(vcfg), 1)
    ren = dict(zip(("to_q", "to_k", "to_v", "to_out.0"), vae_attn_names))
    vsd_disk = {}
    for k, v in vsd.items():
        for new, old in ren.items():
            k = k.replace(f".attentions.0.{new}.", f".attentions.0.{old}.")
        vsd_disk[k] = v
    vsd_disk["encoder.conv_in.weight"] = torch.zeros(4, 3, 3, 3, dtype=torch.float16)     # must be ignored
    save_file(usd, os.path.join(root, "unet", "diffusion_pytorch_model.safetensors"))
    save_file(vsd_disk, os.path.join(root, "vae", "diffusion_pytorch_model.safetensors"))
    full_u = unet_config(ucfg)
    json.dump({"block_out_channels": list(full_u["block_out_channels"]), "cross_attention_dim": full_u["cross_attention_dim"],
               "attention_head_dim": full_u["attention_head_dim"], "norm_num_groups": full_u["norm_num_groups"],
               "time_cond_proj_dim": full_u["time_cond_proj_dim"], "layers_per_block": 2, "in_channels": 4, "out_channels": 4,
               "down_block_types": ["CrossAttnDownBlock2D"] * 3 + ["DownBlock2D"]},
              open(os.path.join(root, "unet", "config.json"), "w"))
    full_v = vae_config(vcfg)
    json.dump({"block_out_channels": list(full_v["block_out_channels"]), "norm_num_groups": full_v["norm_num_groups"],
               "scaling_factor": 0.18215, "latent_channels": 4, "layers_per_block": 2},
              open(os.path.join(root, "vae", "config.json"), "w"))
    json.dump({"_class_name": "StableDiffusionPipeline"}, open(os.path.join(root, "model_index.json"), "w"))
    json.dump({"_class_name": "PNDMScheduler", "beta_start": 0.00085, "beta_end": 0.012, "beta_schedule": "scaled_linear",
               "num_train_timesteps": 1000, "skip_prk_steps": True}, open(os.path.join(root, "scheduler", "scheduler_config.json"), "w"))
    return usd, vsd


UCFG = dict(block_out_channels=(64, 128, 128, 128), attention_head_dim=8, cross_attention_dim=768, norm_num_groups=32,
            time_cond_proj_dim=256)
VCFG = dict(block_out_channels=(64, 64, 128, 128), norm_num_groups=32)


@pytest.mark.parametrize("legacy_names", [False, True])
def test_diffusers_dir_round_trip(tmp_path, legacy_names):
    from sdlcm_amd import weights
    from sdlcm_amd.scheduler import LCMSchedule
    root = str(tmp_path / "model")
    names = ("query", "key", "value", "proj_attn") if legacy_names else ("to_q", "to_k", "to_v", "to_out.0")
    usd, vsd = _write_ckpt(root, UCFG, VCFG, names)
    lu, lucfg, lv, lvcfg = weights.load_diffusers_dir(root)
    assert lucfg["block_out_channels"] == (64, 128, 128, 128) and lucfg["time_cond_proj_dim"] == 256
    assert lucfg["down_attn"] == (True, True, True, False)
    assert set(lu) == set(usd) and all(torch.equal(lu[k], usd[k]) for k in usd)
    assert set(lv) == set(vsd) and all(torch.equal(lv[k], vsd[k]) for k in vsd)          # encoder.* dropped, names mapped
    # LCMScheduler.from_config semantics: foreign scheduler keys are ignored, shared ones kept
    s = LCMSchedule.from_config_file(os.path.join(root, "scheduler", "scheduler_config.json"))
    assert list(s.timesteps(4)) == [999, 759, 499, 259]


def test_loader_rejects_graph_mismatch(tmp_path):
    from safetensors.torch import load_file, save_file
    from sdlcm_amd import weights
    root = str(tmp_path / "model")
    _write_ckpt(root, UCFG, VCFG)
    p = os.path.join(root, "unet", "diffusion_pytorch_model.safetensors")
    sd = load_file(p)
    del sd["mid_block.attentions.0.proj_out.weight"]
    save_file(sd, p)
    with pytest.raises(RuntimeError, match="checkpoint/graph mismatch at unet 'mid_block.attentions.0.proj_out.weight'"):
        weights.load_diffusers_dir(root)


def test_factory_detects_family_from_checkpoint(tmp_path, monkeypatch):
    from sdlcm_amd.backends import worker_factory
    root = str(tmp_path / "model")
    _write_ckpt(root, UCFG, VCFG)
    monkeypatch.setenv("MODEL_ROOT", str(tmp_path))
    monkeypatch.setenv("MODEL", "model")
    assert worker_factory.detect_worker_type() == "sd15"
    cfgp = os.path.join(root, "unet", "config.json")
    j = json.load(open(cfgp))
    j["cross_attention_dim"] = 2048
    json.dump(j, open(cfgp, "w"))
    assert worker_factory.detect_worker_type() == "sdxl"
    j["cross_attention_dim"] = 999
    json.dump(j, open(cfgp, "w"))
    with pytest.raises(RuntimeError, match="Unknown cross_attention_dim"):
        worker_factory.detect_worker_type()


@pytest.mark.gpu
def test_worker_runs_from_checkpoint_directory(tmp_path, monkeypatch):
    """A (narrow, synthetic) diffusers-layout checkpoint drives the real worker end to end on the MI355X."""
    from sdlcm_amd.backends import worker_factory
    from dataclasses import dataclass

    @dataclass
    class Req:
        prompt: str = "a lighthouse"
        size: str = "128x128"
        num_inference_steps: int = 2
        guidance_scale: float = 1.0
        seed: int = 3

    @dataclass
    class J:
        req: Req

    root = str(tmp_path / "model")
    # 2 heads x 64 = head_dim 64 at every level (the attention kernel is instantiated for 40/64/80/160)
    _write_ckpt(root, dict(UCFG, block_out_channels=(128, 128, 128, 128), attention_head_dim=2),
                dict(VCFG, block_out_channels=(64, 128, 128, 128)))
    monkeypatch.setenv("MODEL_ROOT", str(tmp_path))
    monkeypatch.setenv("MODEL", "model")
    monkeypatch.delenv("LCM_HIP_SYNTHETIC", raising=False)
    w = worker_factory.create_hip_worker(worker_id=3)
    try:
        png, seed = w.run_job(J(Req()))
        assert w.worker_id == 3 and seed == 3 and png[:8] == b"\x89PNG\r\n\x1a\n"
        png2, _, lat = w.run_job_with_latents(J(Req()))
        assert png2 == png and len(lat) == 512
    finally:
        w.close()


def _to_ldm_names(usd, vsd, csd, up_attn=(False, True, True, True)):
    """Independent inverse of the loader's mapping: diffusers names -> original (LDM) single-file names."""
    out = {}
    res = {"norm1": "in_layers.0", "conv1": "in_layers.2", "time_emb_proj": "emb_layers.1", "norm2": "out_layers.0",
           "conv2": "out_layers.3", "conv_shortcut": "skip_connection"}

    def r(rest):
        head, tail = rest.split(".", 1)
        return res[head] + "." + tail

    for k, v in usd.items():
        p = k.split(".")
        if p[0] == "conv_in":
            n = "input_blocks.0.0." + p[1]
        elif p[0] == "time_embedding":
            n = "time_embed.cond_proj.weight" if p[1] == "cond_proj" else f"time_embed.{0 if p[1] == 'linear_1' else 2}.{p[2]}"
        elif p[0] == "add_embedding":
            n = f"label_emb.0.{0 if p[1] == 'linear_1' else 2}.{p[2]}"
        elif p[0] == "down_blocks":
            b = int(p[1])
            if p[2] == "downsamplers":
                n = f"input_blocks.{3 * b + 3}.0.op.{p[-1]}"
            else:
                i = 3 * b + 1 + int(p[3])
                n = f"input_blocks.{i}.0." + r(".".join(p[4:])) if p[2] == "resnets" else f"input_blocks.{i}.1." + ".".join(p[4:])
        elif p[0] == "mid_block":
            n = f"middle_block.{2 * int(p[2])}." + r(".".join(p[3:])) if p[1] == "resnets" else "middle_block.1." + ".".join(p[3:])
        elif p[0] == "up_blocks":
            b = int(p[1])
            if p[2] == "upsamplers":
                n = f"output_blocks.{3 * b + 2}.{2 if up_attn[b] else 1}.conv.{p[-1]}"
            else:
                i = 3 * b + int(p[3])
                n = f"output_blocks.{i}.0." + r(".".join(p[4:])) if p[2] == "resnets" else f"output_blocks.{i}.1." + ".".join(p[4:])
        elif p[0] == "conv_norm_out":
            n = "out.0." + p[1]
        else:
            n = "out.2." + p[1]
        out["model.diffusion_model." + n] = v
    for k, v in vsd.items():
        if k.startswith("post_quant_conv."):
            n = k
        else:
            p = k.split(".")[1:]
            if p[0] in ("conv_in", "conv_out"):
                n = "decoder." + ".".join(p)
            elif p[0] == "conv_norm_out":
                n = "decoder.norm_out." + p[1]
            elif p[0] == "mid_block" and p[1] == "resnets":
                n = f"decoder.mid.block_{int(p[2]) + 1}." + ".".join(p[3:]).replace("conv_shortcut", "nin_shortcut")
            elif p[0] == "mid_block":
                a = ".".join(p[3:-1])
                n = "decoder.mid.attn_1." + {"group_norm": "norm", "to_q": "q", "to_k": "k", "to_v": "v", "to_out.0": "proj_out"}[a] + "." + p[-1]
                if a != "group_norm" and p[-1] == "weight":
                    v = v.reshape(v.shape[0], v.shape[1], 1, 1)           # LDM stores the attention projections as 1x1 convs
            elif p[2] == "resnets":
                n = f"decoder.up.{3 - int(p[1])}.block.{p[3]}." + ".".join(p[4:]).replace("conv_shortcut", "nin_shortcut")
            else:
                n = f"decoder.up.{3 - int(p[1])}.upsample.conv.{p[-1]}"
        out["first_stage_model." + n] = v
    for k, v in (csd or {}).items():
        out["cond_stage_model.transformer.text_model." + k] = v
    if csd:
        out["cond_stage_model.transformer.text_model.embeddings.position_ids"] = torch.arange(77).unsqueeze(0)
    out["first_stage_model.encoder.conv_in.weight"] = torch.zeros(4, 3, 3, 3, dtype=torch.float16)
    return out


def _to_openclip_names(sd):
    """transformers CLIPTextModelWithProjection names -> the OpenCLIP text-tower names SDXL single files use."""
    out, L = {}, 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("encoder.layers."))
    out["token_embedding.weight"] = sd["embeddings.token_embedding.weight"]
    out["positional_embedding"] = sd["embeddings.position_embedding.weight"]
    for t in ("weight", "bias"):
        out[f"ln_final.{t}"] = sd[f"final_layer_norm.{t}"]
    out["text_projection"] = sd["text_projection.weight"].t().contiguous()
    out["logit_scale"] = torch.tensor(4.6)
    for i in range(L):
        a, b = f"encoder.layers.{i}.", f"transformer.resblocks.{i}."
        for t in ("weight", "bias"):
            out[b + f"ln_1.{t}"], out[b + f"ln_2.{t}"] = sd[a + f"layer_norm1.{t}"], sd[a + f"layer_norm2.{t}"]
            out[b + f"mlp.c_fc.{t}"], out[b + f"mlp.c_proj.{t}"] = sd[a + f"mlp.fc1.{t}"], sd[a + f"mlp.fc2.{t}"]
            out[b + f"attn.out_proj.{t}"] = sd[a + f"self_attn.out_proj.{t}"]
            out[b + f"attn.in_proj_{t}"] = torch.cat([sd[a + f"self_attn.{n}_proj.{t}"] for n in "qkv"], dim=0)
    return out


def test_sdxl_single_file_checkpoint_round_trip(tmp_path, monkeypatch):
    """Original-layout SDXL .safetensors: 3-level UNet with label_emb, CLIP-L in transformers names under
    conditioner.embedders.0, OpenCLIP bigG names under conditioner.embedders.1 (fused in_proj, transposed projection)."""
    from safetensors.torch import save_file
    from sdlcm_amd import weights
    from sdlcm_amd.clip import clip_param_spec, synthetic_clip
    from sdlcm_amd.config import SDXL_UNET, unet_config
    from sdlcm_amd.backends import worker_factory
    ucfg = unet_config(dict(SDXL_UNET, block_out_channels=(64, 128, 192), attention_head_dim=(1, 2, 3), cross_attention_dim=1280,
                            transformer_layers_per_block=(1, 2, 3), projection_class_embeddings_input_dim=256 + 6 * 256))
    usd = weights.synthetic_state_dict(weights.unet_param_spec(ucfg), 0)
    vsd = weights.synthetic_state_dict(weights.vae_param_spec(VCFG), 1)
    c1 = synthetic_clip(dict(num_hidden_layers=1, hidden_size=512, intermediate_size=1024, num_attention_heads=8, vocab_size=1000))
    c2 = synthetic_clip(dict(num_hidden_layers=2, hidden_size=768, intermediate_size=1536, num_attention_heads=12, vocab_size=1000,
                             hidden_act="gelu", projection_dim=256), seed=3)
    assert "text_projection.weight" in c2
    raw = _to_ldm_names(usd, vsd, None, up_attn=(True, True, False))
    for k, v in c1.items():
        raw["conditioner.embedders.0.transformer.text_model." + k] = v
    for k, v in _to_openclip_names(c2).items():
        raw["conditioner.embedders.1.model." + k] = v
    path = str(tmp_path / "sdxl.safetensors")
    save_file({k: v.contiguous() for k, v in raw.items()}, path)
    lu, lucfg, lv, lvcfg, (l1, l2) = weights.load_single_file_sdxl(path)
    assert lucfg["block_out_channels"] == (64, 128, 192) and lucfg["cross_attention_dim"] == 1280
    assert tuple(lucfg["transformer_layers_per_block"]) == (1, 2, 3) and lucfg["projection_class_embeddings_input_dim"] == 1792
    assert lvcfg["scaling_factor"] == 0.13025 and lvcfg["sample_size"] == 1024
    for got, ref in ((lu, usd), (lv, vsd), (l1, c1), (l2, c2)):
        assert set(got) == set(ref), set(got) ^ set(ref)
        assert all(torch.equal(got[k], ref[k]) for k in ref)
    monkeypatch.setenv("MODEL_ROOT", str(tmp_path))
    monkeypatch.setenv("MODEL", "sdxl.safetensors")
    assert worker_factory.detect_worker_type() == "sdxl"
    with pytest.raises(RuntimeError):
        weights.load_single_file_sdxl(str(tmp_path / "model_sd15_missing.safetensors")) if False else weights.load_single_file_sdxl(_sd15_file(tmp_path))


@pytest.mark.gpu
def test_sdxl_worker_runs_from_single_file(tmp_path, monkeypatch):
    """A narrow synthetic SDXL single-file checkpoint (LDM UNet names, CLIP-L + OpenCLIP text towers inside) drives
    HipLcmSDXLWorker end to end, with and without classifier-free guidance."""
    from dataclasses import dataclass
    from safetensors.torch import save_file
    from sdlcm_amd import weights
    from sdlcm_amd.clip import synthetic_clip
    from sdlcm_amd.config import SDXL_UNET, unet_config
    from sdlcm_amd.backends import worker_factory

    @dataclass
    class Req:
        prompt: str = "a lighthouse"
        size: str = "128x128"
        num_inference_steps: int = 2
        guidance_scale: float = 1.0
        seed: int = 3

    @dataclass
    class J:
        req: Req

    ucfg = unet_config(dict(SDXL_UNET, block_out_channels=(64, 128, 192), attention_head_dim=(1, 2, 3), cross_attention_dim=1280,
                            transformer_layers_per_block=(1, 2, 3), projection_class_embeddings_input_dim=256 + 6 * 256))
    usd = weights.synthetic_state_dict(weights.unet_param_spec(ucfg), 0)
    vsd = weights.synthetic_state_dict(weights.vae_param_spec(dict(VCFG, block_out_channels=(64, 128, 128, 128))), 1)
    c1 = synthetic_clip(dict(num_hidden_layers=1, hidden_size=512, intermediate_size=1024, num_attention_heads=8, vocab_size=1000))
    c2 = synthetic_clip(dict(num_hidden_layers=2, hidden_size=768, intermediate_size=1536, num_attention_heads=12, vocab_size=1000,
                             hidden_act="gelu", projection_dim=256), seed=3)
    raw = _to_ldm_names(usd, vsd, None, up_attn=(True, True, False))
    raw.update({"conditioner.embedders.0.transformer.text_model." + k: v for k, v in c1.items()})
    raw.update({"conditioner.embedders.1.model." + k: v for k, v in _to_openclip_names(c2).items()})
    save_file({k: v.contiguous() for k, v in raw.items()}, str(tmp_path / "sdxl.safetensors"))
    monkeypatch.setenv("MODEL_ROOT", str(tmp_path))
    monkeypatch.setenv("MODEL", "sdxl.safetensors")
    monkeypatch.delenv("LCM_HIP_SYNTHETIC", raising=False)
    # a single-file checkpoint carries no vocabulary: real text-encoder weights without one are refused, never hash-tokenised
    monkeypatch.delenv("LCM_TOKENIZER_DIR", raising=False)
    with pytest.raises(RuntimeError, match="LCM_TOKENIZER_DIR"):
        worker_factory.create_hip_worker(worker_id=5)
    import tinytok
    tinytok.write(str(tmp_path / "vocab" / "tokenizer"))
    tinytok.write(str(tmp_path / "vocab" / "tokenizer_2"), pad="!")
    monkeypatch.setenv("LCM_TOKENIZER_DIR", str(tmp_path / "vocab"))
    w = worker_factory.create_hip_worker(worker_id=5)
    try:
        assert type(w).__name__ == "HipLcmSDXLWorker"
        png, seed = w.run_job(J(Req()))
        assert seed == 3 and png[:8] == b"\x89PNG\r\n\x1a\n"
        png2, _ = w.run_job(J(Req()))
        assert png2 == png
        png3, _ = w.run_job(J(Req(guidance_scale=5.0)))
        assert png3[:8] == b"\x89PNG\r\n\x1a\n" and png3 != png
    finally:
        w.close()


def _sd15_file(tmp_path):
    from safetensors.torch import save_file
    from sdlcm_amd import weights
    usd = weights.synthetic_state_dict(weights.unet_param_spec(dict(UCFG, block_out_channels=(64, 128, 192, 192))), 0)
    vsd = weights.synthetic_state_dict(weights.vae_param_spec(VCFG), 1)
    p = str(tmp_path / "sd15.safetensors")
    save_file({k: v.contiguous() for k, v in _to_ldm_names(usd, vsd, None).items()}, p)
    return p


def test_single_file_checkpoint_round_trip(tmp_path, monkeypatch):
    """Original-layout single .safetensors (UNet + VAE + CLIP in one file, the format modes.yaml.example points at):
    the LDM -> diffusers key mapping must restore every tensor, infer the architecture, and feed the factory."""
    from safetensors.torch import save_file
    from sdlcm_amd import weights
    from sdlcm_amd.clip import synthetic_clip
    from sdlcm_amd.backends import worker_factory
    ucfg = dict(UCFG, block_out_channels=(64, 128, 192, 192))
    usd = weights.synthetic_state_dict(weights.unet_param_spec(ucfg), 0)
    vsd = weights.synthetic_state_dict(weights.vae_param_spec(VCFG), 1)
    ccfg = dict(num_hidden_layers=2, hidden_size=128, intermediate_size=256, num_attention_heads=2, vocab_size=1000)
    csd = synthetic_clip(ccfg)
    path = str(tmp_path / "model.safetensors")
    save_file({k: v.contiguous() for k, v in _to_ldm_names(usd, vsd, csd).items()}, path)
    lu, lucfg, lv, lvcfg, lc = weights.load_single_file(path)
    assert lucfg["block_out_channels"] == (64, 128, 192, 192) and lucfg["cross_attention_dim"] == 768
    assert lucfg["time_cond_proj_dim"] == 256 and lvcfg["block_out_channels"] == (64, 64, 128, 128)
    assert set(lu) == set(usd) and all(torch.equal(lu[k], usd[k]) for k in usd)
    assert set(lv) == set(vsd) and all(torch.equal(lv[k], vsd[k]) for k in vsd)
    assert set(lc) == set(csd) and all(torch.equal(lc[k], csd[k]) for k in csd)
    monkeypatch.setenv("MODEL_ROOT", str(tmp_path))
    monkeypatch.setenv("MODEL", "model.safetensors")
    assert worker_factory.detect_worker_type() == "sd15"
