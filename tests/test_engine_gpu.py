"""The `Engine` mirror beyond the ControlNet / UNet pair of test_sampler_gpu.py::test_engine_surface:

  * the CLIP plan route: `trt_check.py:4-13` replayed line for line, and the constructor / forward sequence of the TensorRT
    variant's FrozenCLIPEmbedder (`ldm_trt/modules/encoders/modules.py:112-124,140`), against the CLIP oracle;
  * `controlunet_model_shape_dict` (`Engine.py:72-77`);
  * a captured hipGraph is never replayed after the shared handle was re-planned (sdeo_configure frees the arenas the
    graph's kernels point into): capture, reconfigure through ANOTHER engine, come back, compare with eager."""
import numpy as np
import pytest
import torch

from tests.common import make_inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tiny_rt():
    from stablediffusioneo_amd import spec as S
    from stablediffusioneo_amd.runtime import SdeoRuntime
    rt = SdeoRuntime(S.UNET_TINY, S.VAE_TINY)
    rt.load_synthetic(0)
    return rt


@pytest.fixture()
def tiny_engines(tiny_rt):
    from stablediffusioneo_amd import Engine as E, spec as S
    saved = (E.Engine.unet_config, E.Engine.vae_config, E.Engine.clip_config, E.Engine.weights_source)
    E.Engine.unet_config, E.Engine.vae_config, E.Engine.clip_config = S.UNET_TINY, S.VAE_TINY, S.CLIP_TINY
    E.Engine.weights_source = "synthetic:0"
    E._shared[(torch.cuda.current_device(), S.UNET_TINY, S.VAE_TINY)] = tiny_rt
    yield E
    E.Engine.unet_config, E.Engine.vae_config, E.Engine.clip_config, E.Engine.weights_source = saved


def test_clip_engine_replays_trt_check(tiny_engines):
    E = tiny_engines
    from oracle.clip_oracle import clip_text_forward
    from stablediffusioneo_amd import spec as S
    cfg = S.CLIP_TINY
    # --- trt_check.py:4-13, with the embedding width of the configured text tower
    clip_engine_path = "/data/Projects/StableDiffusionEO/engine/CLIP_work1_float32_opt.plan"
    clip_engine = E.Engine(clip_engine_path)
    clip_engine.load()
    model_feed_dict = clip_engine.clip_model_shape_dict(1, 77, embedding_dim=cfg.width)
    clip_engine.activate()
    clip_engine.allocate_buffers(model_feed_dict)
    clip_engine.get_engine_infor()
    g = torch.Generator().manual_seed(7)
    tokens = torch.randint(low=0, high=cfg.vocab, size=(1, 77), dtype=torch.int32, generator=g)
    outputs_trt = clip_engine.infer({"input_ids": tokens})["last_hidden_state"].clone()
    # ---
    assert list(clip_engine.tensors) == ["input_ids", "last_hidden_state"]
    assert outputs_trt.shape == (1, 77, cfg.width) and outputs_trt.dtype == torch.float32
    sd = S.synth_state_dict(S.param_spec_clip(cfg), 0, S.NS_CLIP)
    want = clip_text_forward(sd, tokens.long(), cfg.heads)
    err = float((outputs_trt.cpu() - want).abs().max() / want.abs().max())
    print(f"[parity] CLIP engine vs oracle: max|err|/scale = {err:.3e}")
    assert err < 2e-2
    # graph-captured infer gives the same tensor (`Engine.py:139-152`)
    a = clip_engine.infer({"input_ids": tokens}, use_cuda_graph=True)["last_hidden_state"].clone()
    b = clip_engine.infer({"input_ids": tokens}, use_cuda_graph=True)["last_hidden_state"].clone()
    assert torch.equal(a, outputs_trt) and torch.equal(b, outputs_trt)


def test_clip_engine_in_the_trt_embedder_sequence(tiny_engines):
    """`ldm_trt/modules/encoders/modules.py:112-124` (constructor) and `:134-140` (forward): tokens.int() -> infer -> clone."""
    E = tiny_engines
    from stablediffusioneo_amd import spec as S
    from stablediffusioneo_amd.ldm.modules.encoders.modules import HashTokenizer
    cfg = S.CLIP_TINY
    max_length = 77
    eng = E.Engine("/data/Projects/StableDiffusionEO/engine/CLIP.plan")
    eng.load()
    model_feed_dict = eng.clip_model_shape_dict(1, max_length, embedding_dim=cfg.width)
    eng.activate()
    eng.allocate_buffers(model_feed_dict)
    eng.get_engine_infor()
    tok = HashTokenizer(cfg.vocab, max_length)
    outs = []
    for text in ("a bird, best quality", "lowres, bad anatomy"):
        tokens = torch.from_numpy(tok([text])).to("cuda")
        tokens = tokens.int()
        outs.append(eng.infer({"input_ids": tokens})["last_hidden_state"].clone())
    assert outs[0].shape == (1, max_length, cfg.width) and not torch.equal(outs[0], outs[1])
    direct = eng.engine.encode(torch.from_numpy(tok(["a bird, best quality"])))
    assert torch.equal(direct, outs[0])


def test_controlunet_model_shape_dict(tiny_engines):
    e = tiny_engines.Engine("/x/ControlledUnet.plan")
    e.batch_size, e.latent_h, e.latent_w = 1, 32, 48
    assert e.controlunet_model_shape_dict() == {"sample": (2, 4, 32, 48), "encoder_hidden_states": (2, 77, 768),
                                                "latent": (2, 4, 32, 48)}
    with pytest.raises(Exception, match="engine name"):
        tiny_engines.Engine("/x/yolov5.plan").load()


def test_frozen_clip_embedder_needs_a_real_tokenizer():
    from stablediffusioneo_amd import spec as S
    from stablediffusioneo_amd.ldm.modules.encoders.modules import FrozenCLIPEmbedder
    with pytest.raises(RuntimeError, match="tokenizer"):
        FrozenCLIPEmbedder(version="openai/clip-vit-large-patch14", config=S.CLIP_TINY)


def test_captured_graph_is_dropped_when_the_handle_is_replanned(tiny_engines, tiny_rt):
    E = tiny_engines
    from stablediffusioneo_amd import spec as S
    un = E.Engine("/data/engine/ControlledUnet.plan")
    un.load(); un.activate()
    un.allocate_buffers({"x_noisy": (1, 4, 8, 8)})
    x, ctx, hint = make_inputs(1, 8, 8, ctx_dim=S.UNET_TINY.context_dim)
    ts = torch.full((1,), 401, dtype=torch.long)
    ctrl = tiny_rt.configure(1, 8, 8).controlnet(x, hint, ts, ctx)
    feed = {"x_noisy": x, "timestep": ts, "context": ctx}
    feed.update({f"control{i}": c for i, c in enumerate(ctrl)})
    eager = un.infer(feed)["latent"].clone()
    g1 = un.infer(feed, use_cuda_graph=True)["latent"].clone()        # eager run + capture
    g2 = un.infer(feed, use_cuda_graph=True)["latent"].clone()        # replay
    assert torch.equal(eager, g1) and torch.equal(eager, g2)
    gen = tiny_rt.generation
    # another engine re-plans the shared handle (sdeo_configure frees and re-allocates the arenas) ...
    other = E.Engine("/data/engine/ControlNet.plan")
    other.load(); other.activate()
    other.allocate_buffers({"x_noisy": (2, 4, 16, 8)})
    x2, ctx2, hint2 = make_inputs(2, 16, 8, ctx_dim=S.UNET_TINY.context_dim)
    other.infer({"x_noisy": x2, "hint": hint2, "timestep": torch.tensor([1, 801]), "context": ctx2})
    assert tiny_rt.generation > gen and (tiny_rt.n, tiny_rt.h, tiny_rt.w) == (2, 16, 8)
    # ... and the first engine must re-bind and re-capture instead of replaying into freed memory
    g3 = un.infer(feed, use_cuda_graph=True)["latent"].clone()
    g4 = un.infer(feed, use_cuda_graph=True)["latent"].clone()
    assert torch.equal(eager, g3) and torch.equal(eager, g4)
    assert un._graph_generation == tiny_rt.generation
    # same engine re-allocated to another shape and back
    un.allocate_buffers({"x_noisy": (1, 4, 8, 16)})
    assert un.cuda_graph_instance is None
    un.allocate_buffers({"x_noisy": (1, 4, 8, 8)})
    g5 = un.infer(feed, use_cuda_graph=True)["latent"].clone()
    assert torch.equal(eager, g5)
    torch.cuda.synchronize()
    assert np.isfinite(g5.cpu().numpy()).all()
