#!/usr/bin/env python3
"""Behavioural fixtures of the worker glue, recorded from the reference itself (build container only; needs /root/reference).

Under the same ``sys.modules`` stubs as make_golden.py (diffusers / rknnlite are absent; the reference's own tests stub them the
same way, tests/test_worker_pool.py:14-17) this drives

  DiffusersCudaWorker.run_job            backends/cuda_worker.py:201-239   with a recording fake ``pipe``: the keyword arguments it
                                         receives, the generator's seed, the order of the ``_apply_style`` calls around it (the
                                         reference's own ``_apply_style`` runs: what reaches ``pipe.set_adapters`` is recorded),
                                         the seed policy and the error text for malformed sizes
  RKNN2LatentConsistencyPipeline.check_inputs   backends/rknnlcm.py:370-415   accept / reject cases with the messages
  rknn_worker._latent_to_nchw            backends/rknn_worker.py:182-220   layouts in -> NCHW out, and the error cases
  WorkerPool                             backends/worker_pool.py:135-419   the scripted session of tests/pool_scenario.py with a
                                         recording fake factory / worker: factory keywords, thread names, environment, futures,
                                         exception path, same-mode switch no-op, teardown order, queue-full error

Output: tests/golden/worker_contract.json (+ the arrays in tests/golden/worker_contract.npz).  Data only: inputs, recorded calls,
outputs, messages -- no reference source text.
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference  # noqa: E402  (registers the stubs, puts /root/reference on sys.path)


class Req:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class Job:
    def __init__(self, req):
        self.req = req


class SL:
    def __init__(self, style, level):
        self.style, self.level = style, level


def record_run_job():
    import backends.cuda_worker as cw
    from PIL import Image
    events = []

    class FakePipe:
        def __call__(self, **kw):
            g = kw.pop("generator")
            events.append(["pipe", dict(kw, generator_initial_seed=int(g.initial_seed()), generator_device=str(g.device))])
            return Req(images=[Image.new("RGB", (8, 8), (1, 2, 3))])

        def set_adapters(self, names, adapter_weights=None):
            events.append(["set_adapters", list(names), [float(w) for w in (adapter_weights or [])]])

        def disable_lora(self):
            events.append(["disable_lora"])

    W = cw.DiffusersCudaWorker
    w = W.__new__(W)
    w.pipe, w.device, w.worker_id = FakePipe(), "cpu", 0
    w._style_loaded = {sd.adapter_name: True for sd in cw.STYLE_REGISTRY.values()}
    w._style_api = "adapters"
    cases = [
        dict(prompt="a cat", size="512x512", num_inference_steps=4, guidance_scale=1.0, seed=42),
        dict(prompt="a dog", size="640X360", num_inference_steps=8, guidance_scale=7.5, seed=0),
        dict(prompt="no seed", size="256x256", num_inference_steps=2, guidance_scale=1.0, seed=None),
        dict(prompt="styled", size="512x512", num_inference_steps=4, guidance_scale=1.0, seed=7, style_lora=("papercut", 2)),
        dict(prompt="styled hi", size="512x512", num_inference_steps=4, guidance_scale=1.0, seed=8, style_lora=("papercut", 9)),
        dict(prompt="style off", size="512x512", num_inference_steps=4, guidance_scale=1.0, seed=9, style_lora=("papercut", 0)),
        dict(prompt="unknown style", size="512x512", num_inference_steps=4, guidance_scale=1.0, seed=10, style_lora=("nope", 2)),
        dict(prompt="bad size", size="512", num_inference_steps=4, guidance_scale=1.0, seed=1),
        dict(prompt="bad size 2", size="axb", num_inference_steps=4, guidance_scale=1.0, seed=1),
        dict(prompt="steps as str", size="128x64", num_inference_steps="3", guidance_scale="2", seed="5"),
    ]
    out = []
    for c in cases:
        del events[:]
        kw = dict(c)
        sl = kw.pop("style_lora", None)
        req = Req(**kw)
        if sl is not None:
            req.style_lora = SL(*sl)
        rec = dict(request=dict(c))
        try:
            png, seed = w.run_job(Job(req))
            rec["png_magic_ok"] = bool(png[:8] == b"\x89PNG\r\n\x1a\n")
            rec["returned_seed"] = int(seed)
            rec["seed_is_request_seed"] = c["seed"] is not None and int(seed) == int(c["seed"])
        except Exception as e:      # noqa
            rec["error_type"], rec["error"] = type(e).__name__, str(e)
        rec["events"] = [list(e) for e in events]
        out.append(rec)
    # seed policy without a seed: range of the reference's draw
    seeds = []
    for _ in range(64):
        del events[:]
        _, s = w.run_job(Job(Req(prompt="p", size="64x64", num_inference_steps=1, guidance_scale=1.0, seed=None)))
        seeds.append(int(s))
    return out, dict(min=min(seeds), max=max(seeds), distinct=len(set(seeds)), upper_bound_exclusive=100_000_000)


def record_check_inputs(rknnlcm):
    P = rknnlcm.RKNN2LatentConsistencyPipeline
    pipe = P.__new__(P)
    e1, e2 = np.zeros((1, 77, 768), np.float32), np.zeros((2, 77, 768), np.float32)
    cases = [
        dict(prompt="a", height=512, width=512, callback_steps=1),
        dict(prompt="a", height=360, width=640, callback_steps=1),
        dict(prompt="a", height=8, width=8, callback_steps=3),
        dict(prompt="a", height=513, width=512, callback_steps=1),
        dict(prompt="a", height=512, width=100, callback_steps=1),
        dict(prompt="a", height=512, width=512, callback_steps=0),
        dict(prompt="a", height=512, width=512, callback_steps=None),
        dict(prompt="a", height=512, width=512, callback_steps=1.5),
        dict(prompt=["a", "b"], height=512, width=512, callback_steps=1),
        dict(prompt=None, height=512, width=512, callback_steps=1),
        dict(prompt=7, height=512, width=512, callback_steps=1),
        dict(prompt="a", height=512, width=512, callback_steps=1, prompt_embeds="e1"),
        dict(prompt=None, height=512, width=512, callback_steps=1, prompt_embeds="e1"),
        dict(prompt=None, height=512, width=512, callback_steps=1, prompt_embeds="e1", negative_prompt_embeds="e2"),
        dict(prompt="a", height=512, width=512, callback_steps=1, negative_prompt="n", negative_prompt_embeds="e1"),
    ]
    out = []
    for c in cases:
        kw = {k: ({"e1": e1, "e2": e2}[v] if isinstance(v, str) and v in ("e1", "e2") else v) for k, v in c.items()}
        rec = dict(args={k: (v if not isinstance(v, float) or v == int(v) else v) for k, v in c.items()})
        try:
            pipe.check_inputs(**kw)
            rec["outcome"] = "ok"
        except Exception as e:      # noqa
            rec["outcome"], rec["error_type"], rec["error"] = "error", type(e).__name__, str(e)
        out.append(rec)
    return out


def record_latent_to_nchw(rknn_worker):
    rng = np.random.RandomState(11)
    arrays, recs = {}, []
    cases = [("nchw", rng.randn(1, 4, 8, 6).astype(np.float32)), ("nhwc", rng.randn(1, 8, 6, 4).astype(np.float32)),
             ("nchw_b2", rng.randn(2, 4, 5, 5).astype(np.float32)), ("c_axis2", rng.randn(1, 8, 4, 6).astype(np.float32)),
             ("torch_nhwc", torch.from_numpy(rng.randn(1, 3, 5, 4).astype(np.float32))),
             ("list_nchw", rng.randn(1, 4, 2, 2).astype(np.float32).tolist()),
             ("ambiguous_4x4", rng.randn(1, 4, 4, 4).astype(np.float32))]
    for name, x in cases:
        out = rknn_worker._latent_to_nchw(x)
        arrays["l2n_in_" + name] = np.asarray(x.numpy() if hasattr(x, "numpy") and not isinstance(x, np.ndarray) else x, dtype=np.float32)
        arrays["l2n_out_" + name] = np.asarray(out, dtype=np.float32)
        recs.append(dict(name=name, in_shape=list(arrays["l2n_in_" + name].shape), out_shape=list(out.shape)))
    errs = []
    for name, x in (("none", None), ("3d", np.zeros((4, 8, 8), np.float32)), ("no_4", np.zeros((1, 3, 8, 8), np.float32))):
        try:
            rknn_worker._latent_to_nchw(x)
            errs.append(dict(name=name, outcome="ok"))
        except Exception as e:      # noqa
            errs.append(dict(name=name, outcome="error", error_type=type(e).__name__, error=str(e)))
    return recs, errs, arrays


def record_pool():
    """The reference's WorkerPool (backends/worker_pool.py:135-419) driven by tests/pool_scenario.py with a recording fake
    worker, plain stand-ins for its ModeConfigManager / ModelRegistry collaborators (dependency injection, :147-181)."""
    sys.path.insert(0, os.path.dirname(HERE))
    from pool_scenario import run_scenario
    import backends.worker_pool as wp
    wp.reset_worker_pool()

    class Mode:
        def __init__(self, name, model):
            self.name, self.model, self.model_path, self.loras = name, model, "/models/" + model, []

    class ModeConfig:
        class config:
            model_root = "/models"
        modes = {"mode-a": Mode("mode-a", "a.safetensors"), "mode-b": Mode("mode-b", "b.safetensors")}

        def get_mode(self, name):
            return self.modes[name]

        def get_default_mode(self):
            return "mode-a"

    def make_pool(factory, registry_event, queue_max):
        class Registry:
            def get_used_vram(self):
                return 0

            def register_model(self, name, **kw):
                registry_event("register", name)

            def unregister_model(self, name):
                registry_event("unregister", name)

        return wp.WorkerPool(queue_max=queue_max, worker_factory=factory, mode_config=ModeConfig(), registry=Registry())

    import logging
    logging.disable(logging.CRITICAL)
    try:
        return run_scenario(make_pool, wp.GenerationJob, wp.ModeSwitchJob, wp.CustomJob)
    finally:
        logging.disable(logging.NOTSET)


def main():
    rknnlcm, rknn_worker = import_reference()
    run_job, seed_policy = record_run_job()
    l2n, l2n_err, arrays = record_latent_to_nchw(rknn_worker)
    doc = dict(source="recorded from /root/reference under sys.modules stubs (tests/golden/make_contract_golden.py)",
               run_job=run_job, seed_policy_without_seed=seed_policy, check_inputs=record_check_inputs(rknnlcm),
               latent_to_nchw=l2n, latent_to_nchw_errors=l2n_err, pool=record_pool())
    with open(os.path.join(HERE, "worker_contract.json"), "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True, default=str)
    np.savez_compressed(os.path.join(HERE, "worker_contract.npz"), **arrays)
    print("wrote worker_contract.json / .npz:", len(run_job), "run_job cases,", len(doc["check_inputs"]), "check_inputs cases,", len(l2n), "layouts")
    for r in run_job:
        print(" ", r["request"].get("prompt"), "->", r.get("error") or r["events"])


if __name__ == "__main__":
    main()
