"""bench.py -- headline benchmark of the MI355X-native canny2image hot path (BASELINE.json configs[1]).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic input = ONE IMAGE per GPU:
hint block + cross-attention K/V precompute (once per image), 20 DDIM steps of ControlNet + ControlledUnet on the
fused CFG pair (N=2) with the CFG/DDIM update, VAE decode and the uint8 post-process -- all on the GPU, inputs
resident in HBM.  Images are sharded over ranks by image index (no data-path collective); the final latents are
gathered once with RCCL inside the timed region.  Rank 0 prints ONE JSON line.

Extra objects on that line:
  roofline     -- the conv / GEMM kernel family of one profiled image (the group holding most of the device time): algorithmic FLOPs
                  of its launches / their summed HIP-event durations, against the dense fp16 MFMA peak (2.5 PFLOP/s); `traffic` =
                  HBM-side bytes of one DDIM step from the committed rocprofv3 --pmc passes (profiles/r0N_step_counters.json);
                  the single (kernel, shape) pair with the most device time is kept under `dominant_pair`.
  cpu_baseline -- the oracle (oracle/sd_oracle.py = fp32 PyTorch restatement of the reference path, "port") timed
                  on this host's cores on a bounded sample (N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

os.environ.setdefault("SDEO_PROFILE_DETAIL", "1")     # in-library profiler keys = "kernel | problem shape"

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from stablediffusioneo_amd import spec as S                      # noqa: E402

PEAK_TFLOPS_F16 = 2500.0            # dense fp16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
# algorithmic FLOPs (2*MAC over conv + GEMM + attention matmuls of the reference modules)
FLOP_PER_STEP = {256: 0.486e12, 512: 2.173e12, 768: 5.843e12}
FLOP_VAE = {256: 0.622e12, 512: 2.515e12, 768: 5.754e12}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4, help="timed images per GPU")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--ddim-steps", type=int, default=20)
    ap.add_argument("--scale", type=float, default=9.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--config", default="sd15", choices=["sd15", "tiny"])
    ap.add_argument("--fp8", action="store_true",
                    help="BASELINE configs[4]: UNet / ControlNet matrices packed to fp8 e4m3fn (per-output-channel scales), fp16 activations")
    ap.add_argument("--mx", action="store_true",
                    help="with --fp8: the GEMMs of >= --mx-min-rows rows run block-scaled fp8 x fp8 on v_mfma_scale_f32_16x16x128_f8f6f4 "
                         "(activations packed to e4m3fn + e8m0 block scales in front of each such GEMM)")
    ap.add_argument("--mx-min-rows", type=int, default=2048)
    ap.add_argument("--batch", type=int, default=1, help="images per GPU per step (the CFG pair makes N = 2 x batch)")
    ap.add_argument("--fast-weights", action="store_true",
                    help="draw the synthetic weights with the device generator (seconds faster to start; NOT the parity-tested weights, "
                         "so the reference-golden check of the output is skipped)")
    ap.add_argument("--dump-profile", default="", help="write the full per-(kernel, shape) table of the profiled image to this JSON file")
    return ap.parse_args()


def cpu_baseline(res, ddim_steps, scale):
    """Oracle ("port" of the reference PyTorch path, fp32) on the host cores, bounded samples, synthetic weights:
      (a) the benchmark configuration: ONE apply_model pass (ControlNet + ControlledUnet, N=1) and ONE VAE decode at `res`;
          the reference runs two such passes per DDIM step (`cldm/ddim_hacked.py:190-191`), so
          images/s = 1 / (ddim_steps * 2 * t_pass + t_vae)  -- this is `value`;
      (b) BASELINE configs[0] in full (`canny2image_torch.py` shape: 256x256, 5 DDIM steps, CFG, VAE decode), measured, not
          extrapolated -- reported under `config1`."""
    from oracle import sd_oracle as O
    from tests.common import X_T_SEED, make_hint, make_inputs, randn
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, int(os.environ.get("SDEO_CPU_CORES", "16"))))   # the GPU box grants 16 cores per GPU
    torch.set_num_threads(cores)
    u, v = S.UNET_SD15, S.VAE_SD15
    print(f"[bench] cpu_baseline: generating synthetic fp32 weights on {cores} host threads ...", file=sys.stderr, flush=True)
    su = S.synth_state_dict(S.param_spec_unet(u), 0, S.NS_UNET)
    sc = S.synth_state_dict(S.param_spec_controlnet(u), 0, S.NS_CONTROL)
    sv = S.synth_state_dict(S.param_spec_vae(v), 0, S.NS_VAE)
    up, cp, hc, vp = S.unet_plan(u), S.unet_plan(u, False), S.hint_block_convs(u), S.vae_plan(v)[1]
    h = res // 8
    x, ctx, hint = make_inputs(1, h, h)
    t = torch.tensor([951], dtype=torch.long)
    print("[bench] cpu_baseline: timing one oracle apply_model pass + one VAE decode ...", file=sys.stderr, flush=True)
    with torch.no_grad():
        t0 = time.perf_counter()
        O.apply_model(su, sc, up, cp, hc, x, t, ctx, hint, [1.0] * 13)
        dt = time.perf_counter() - t0
        t0 = time.perf_counter()
        O.decode_first_stage(sv, vp, x * v.scale_factor, v.scale_factor)
        dv = time.perf_counter() - t0
        # configs[0]: the whole reference pipeline shape at 256x256, 5 DDIM steps
        print("[bench] cpu_baseline: config 1 (256x256, 5 DDIM steps + VAE) ...", file=sys.stderr, flush=True)
        h1 = 32
        hint1 = make_hint(1, 8 * h1, 8 * h1)
        c1, c2 = randn((1, 77, u.context_dim), 1), randn((1, 77, u.context_dim), 2)

        def apply_fn(xx, tt, cc):
            return O.apply_model(su, sc, up, cp, hc, xx, tt, cc, hint1, [1.0] * 13)

        t0 = time.perf_counter()
        z, _ = O.ddim_sample(apply_fn, randn((1, 4, h1, h1), X_T_SEED), 5, c1, c2, scale)
        img = O.decode_first_stage(sv, vp, z, v.scale_factor)
        d1 = time.perf_counter() - t0
        assert bool(torch.isfinite(img).all())
    return {"value": 1.0 / (ddim_steps * 2 * dt + dv), "unit": "images/s", "cores": cores, "kind": "port",
            "seconds_per_apply_model_pass": round(dt, 3), "seconds_per_vae_decode": round(dv, 3),
            "sample": f"1 apply_model pass (ControlNet + ControlledUnet, N=1) + 1 VAE decode of the fp32 oracle at {res}x{res}; "
                      f"images/s = 1 / ({ddim_steps} steps x 2 passes x t_pass + t_vae)",
            "config1": {"workload": "BASELINE configs[0]: 256x256, 5 DDIM steps (CFG, 2 passes per step) + VAE decode, fp32 oracle, "
                                    "measured in full", "seconds_per_image": round(d1, 3), "images_per_s": round(1.0 / d1, 5)}}


def spawn_ranks(n):
    """`python bench.py --gpus N` without an external launcher: start N copies of this command, one per GPU, with the
    torch.distributed environment set (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT), BEFORE this process has touched
    a device (it never does: the parent only waits).  Rank 0's stdout (the one JSON line) is passed through; the exit code is the
    worst child's.  Under `python -m torch.distributed.run` the environment is already there and nothing is spawned."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = str(sk.getsockname()[1])
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=port)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this platform (RCCL needs it)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def main():
    a = parse()
    if a.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(a.gpus))
    if os.environ.get("SDEO_BENCH_ECHO_ENV"):      # launcher self-test (tests/test_sharding_cpu.py): report the rank environment, touch no device
        print(json.dumps({k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}),
              file=open(os.environ["SDEO_BENCH_ECHO_ENV"] + "." + os.environ.get("RANK", "0"), "w"))
        return
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from stablediffusioneo_amd.cldm.cldm import ControlLDM
    from stablediffusioneo_amd.cldm.ddim_hacked import DDIMSampler
    from stablediffusioneo_amd.runtime import SdeoRuntime
    from tests.common import X_T_SEED, make_hint, randn

    ucfg, vcfg = (S.UNET_SD15, S.VAE_SD15) if a.config == "sd15" else (S.UNET_TINY, S.VAE_TINY)
    rt = SdeoRuntime(ucfg, vcfg, device=dev, weight_bits=8 if a.fp8 else 16, act_bits=8 if (a.fp8 and a.mx) else 16, mx_min_rows=a.mx_min_rows)
    if a.fast_weights:
        rt.load_synthetic_device(0)
    else:
        rt.load_synthetic(0)      # the CPU-generator weights of the parity tests and of the reference goldens
    model = ControlLDM(rt)
    sampler = DDIMSampler(model)
    h = w = a.res // 8
    B = a.batch
    hint = make_hint(1, a.res, a.res).to(dev).expand(B, -1, -1, -1).contiguous()
    ctx_c = randn((1, ucfg.context_len, ucfg.context_dim), 1).to(dev).expand(B, -1, -1).contiguous()
    ctx_u = randn((1, ucfg.context_len, ucfg.context_dim), 2).to(dev).expand(B, -1, -1).contiguous()
    cond = {"c_concat": [hint], "c_crossattn": [ctx_c]}
    unc = {"c_concat": [hint], "c_crossattn": [ctx_u]}
    loop_ev = []

    # the inputs of every unit are resident in HBM before the timed region starts (x_T drawn on the host from the unit's seed)
    from stablediffusioneo_amd.sharding import unit_index
    x_Ts = {}
    for i in range(max(a.warmup, a.steps)):
        u = unit_index(rank, i, world)
        x_Ts[u] = torch.cat([randn((1, 4, h, w), X_T_SEED + u * B + j) for j in range(B)]).to(dev)

    def one_image(index, timed=True):
        if index not in x_Ts:       # the untimed profiling pass draws its own unit
            x_Ts[index] = torch.cat([randn((1, 4, h, w), X_T_SEED + index * B + j) for j in range(B)]).to(dev)
        x_T = x_Ts[index]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        z, _ = sampler.sample(a.ddim_steps, B, (4, h, w), cond, verbose=False, eta=0.0, unconditional_guidance_scale=a.scale,
                              unconditional_conditioning=unc, x_T=x_T)
        e1.record()
        if timed:
            loop_ev.append((e0, e1))
        img = model.decode_first_stage_uint8(z)
        return z, img

    for i in range(a.warmup):
        one_image(unit_index(rank, i, world), timed=False)
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    zs = []
    for i in range(a.steps):
        z, img = one_image(unit_index(rank, i, world))      # unit (one batch of B images) u -> rank u % world
        zs.append(z)
    zloc = torch.stack(zs).half()                 # (steps, B, 4, h, w): one row per unit, in shard_indices order
    gather_ms = None
    if dist:
        from stablediffusioneo_amd.sharding import gather_latents
        torch.cuda.synchronize()
        tg = time.perf_counter()
        zall = gather_latents(zloc.reshape(a.steps, -1, h, w), world * a.steps)   # the only collective: final latents, 32 KiB / image
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - tg) * 1e3
        assert zall.shape[0] == world * a.steps
        # unit u of the gathered tensor must be the unit this rank produced at step (u - rank) / world
        for i in range(a.steps):
            assert torch.equal(zall[unit_index(rank, i, world)].reshape(zloc[i].shape), zloc[i]), "gather order"
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    per_rank = None
    if dist:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        tall = [torch.zeros_like(tmax) for _ in range(world)]
        dist.all_gather(tall, tmax)                 # bookkeeping only (after the timed region): every rank's own wall time
        per_rank = [round(a.steps * B / float(t.item()), 4) for t in tall]
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    loop_ms = sum(e0.elapsed_time(e1) for e0, e1 in loop_ev) / max(len(loop_ev), 1)
    assert torch.isfinite(zloc.float()).all() and int(img.float().std() > 0), "degenerate output"
    # the benched numerics ARE the parity-tested numerics: image index 0 of the default workload is the trajectory the reference's
    # own DDIMSampler + ControlLDM.apply_model produced in tests/golden/sd15_full.npz (same weights, x_T, hint, contexts)
    golden_check = None
    gpath = os.path.join(ROOT, "tests", "golden", "sd15_full.npz")
    if (rank == 0 and not a.fast_weights and not a.fp8 and a.config == "sd15" and (a.res, a.ddim_steps, a.scale) == (512, 20, 9.0)
            and os.path.exists(gpath)):
        import numpy as np
        zref = torch.tensor(np.load(gpath)["traj64.z"])
        err = (zs[0][:1].float().cpu() - zref).abs()
        golden_check = {"what": "final latent of image 0 vs the reference DDIMSampler + ControlLDM.apply_model golden (traj64.z)",
                        "max_abs_err_over_ref_max": round(float(err.max() / zref.abs().max()), 5),
                        "mean_abs_err_over_ref_max": round(float(err.mean() / zref.abs().max()), 6)}
        assert golden_check["max_abs_err_over_ref_max"] < 5e-2, f"benchmarked output is off the reference golden: {golden_check}"

    fp8_check = None
    fpath = os.path.join(ROOT, "tests", "golden", "sd15_fp8.npz")
    if rank == 0 and a.fp8 and not a.fast_weights and a.config == "sd15" and a.res == 512 and os.path.exists(fpath):
        # distance of ONE pass (N = 2, t = 951) to the reference modules run in fp32 on the same fp8-dequantised weights
        # (tests/golden/make_golden_full.py --only pass64_fp8): what is left is the activation precision (fp16, or block-scaled fp8 with --mx)
        import numpy as np
        from stablediffusioneo_amd import _lib as L
        import ctypes as C
        x1 = randn((1, 4, h, w), X_T_SEED)
        xx = torch.cat([x1, x1]).to(dev)
        hint1 = make_hint(1, a.res, a.res)
        cc = torch.cat([randn((1, 77, 768), 1), randn((1, 77, 768), 2)]).to(dev)
        rt.configure(2, h, w)
        eps = rt.apply_model(xx, torch.cat([hint1, hint1]).to(dev), torch.tensor([951, 951], dtype=torch.long, device=dev), cc, scales=[1.0] * 13)
        ref = torch.tensor(np.load(fpath)["pass64_fp8.eps"])
        err = (eps.float().cpu() - ref).abs()
        L.load().sdeo_debug_mx_launches.argtypes = [C.c_void_p]
        fp8_check = {"what": "eps of one N=2 pass vs the reference modules (fp32) on the same fp8-dequantised weights (pass64_fp8.eps)",
                     "max_abs_err_over_ref_max": round(float(err.max() / ref.abs().max()), 5),
                     "mean_abs_err_over_ref_max": round(float(err.mean() / ref.abs().max()), 6),
                     "gemm_launches_on_fp8_mfma_in_configured_programs": int(L.load().sdeo_debug_mx_launches(rt.handle))}      # ControlNet + UNet (+ the no-control UNet variant)
    roof = None
    if not a.no_roofline and rank == 0:
        import stablediffusioneo_amd.cldm.ddim_hacked as dh
        dh.USE_GRAPH = False          # events are recorded around eager launches only (a graph replay has no host-side hooks)
        rt.profile_begin()
        one_image(10_000, timed=False)
        prof = rt.profile_end()
        dh.USE_GRAPH = True
        # records are per (kernel, problem shape); the roofline object describes the pair with the largest share of device time
        mm = [k for k in prof if k["flops"] > 0]
        dom = max(mm, key=lambda k: k["total_ms"])
        tot_ms = sum(k["total_ms"] for k in prof)
        ach = dom["flops"] / (dom["total_ms"] * 1e-3) / 1e12
        by_kernel, fam = {}, {}
        for k in prof:
            name = k["kernel"].split(" | ")[0]
            by_kernel[name] = by_kernel.get(name, 0.0) + k["total_ms"]
            f = ("conv_gemm" if name.startswith(("conv_gemm", "conv3x3_halo")) else name)
            e = fam.setdefault(f, {"ms": 0.0, "flops": 0.0, "launches": 0})
            e["ms"] += k["total_ms"]; e["flops"] += k["flops"]; e["launches"] += k["launches"]
        families = {f: {"ms_per_image": round(e["ms"], 2), "launches_per_image": e["launches"],
                        "tflops": round(e["flops"] / (e["ms"] * 1e-3) / 1e12, 1) if e["flops"] else None,
                        "frac_of_mfma_peak": round(e["flops"] / (e["ms"] * 1e-3) / 1e12 / PEAK_TFLOPS_F16, 4) if e["flops"] else None,
                        "share_of_device_time": round(e["ms"] / tot_ms, 3)} for f, e in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])}
        def committed(*names):           # newest committed rocprofv3 result of that kind (PMC numbers cannot be taken in this process)
            for nm in names:
                pth = os.path.join(ROOT, "profiles", nm)
                if os.path.exists(pth):
                    return json.load(open(pth)), nm
            return None, None
        tj, tname = committed("r03_traffic.json", "r02_traffic.json", "r01_traffic.json")
        pair_traffic, tnote = None, None
        if tj and tj.get(dom["kernel"]):
            pair_traffic, tnote = tj[dom["kernel"]]["traffic_bytes"], tj[dom["kernel"]]["shape"] + f" ({tname})"
        sj, sname = committed("r03_step_counters.json", "r02_step_counters.json")
        step_counters = sj.get("summary") if sj else None
        step_traffic = None
        if step_counters:          # HBM-side bytes of ONE DDIM step (FETCH_SIZE x 2 + WRITE_SIZE, eager single-stream counter pass)
            step_traffic = int((step_counters.get("l2_fabric_read_GB_per_step", 0) + step_counters.get("l2_fabric_write_GB_per_step", 0)) * 1e9) or None
        # The headline fraction is that of the conv / GEMM FAMILY (all implicit-GEMM, halo-conv and GEMM launches of the image: the
        # group that holds most of the device time), not of the single best (kernel, shape) pair -- the pair is kept under
        # `dominant_pair`.  achieved = sum of algorithmic FLOP of the family's launches / sum of their HIP-event durations.
        cg = fam["conv_gemm"]
        cg_ach = cg["flops"] / (cg["ms"] * 1e-3) / 1e12
        roof = {"bound": "mfma", "achieved": round(cg_ach, 2), "peak": PEAK_TFLOPS_F16, "unit": "TFLOP/s",
                "frac": round(cg_ach / PEAK_TFLOPS_F16, 4),
                "traffic": step_traffic,
                "traffic_is": f"HBM-side bytes of one DDIM step, all kernels (FETCH_SIZE x 2 + WRITE_SIZE; {sname}); per-kernel passes: profiles/r03_gemm_counters.json",
                "kernel": "conv / GEMM family: conv_gemm_dma_kernel<*> + conv3x3_halo_kernel<*> + conv_gemm_kernel<*> (every F.conv2d / F.linear of the path)",
                "launches_per_image": cg["launches"], "avg_launch_us": round(cg["ms"] * 1e3 / cg["launches"], 2),
                "share_of_device_time": round(cg["ms"] / tot_ms, 3),
                "dominant_pair": {"kernel": dom["kernel"], "achieved": round(ach, 2), "frac": round(ach / PEAK_TFLOPS_F16, 4),
                                  "launches_per_image": dom["launches"], "avg_launch_us": round(dom["total_ms"] * 1e3 / dom["launches"], 2),
                                  "share_of_device_time": round(dom["total_ms"] / tot_ms, 3),
                                  "traffic": pair_traffic, "traffic_measured_on": tnote},
                "launches_per_image_all_kernels": sum(k["launches"] for k in prof),
                "families": families,
                "whole_step_counters": step_counters,
                "top_shapes_ms_per_image": {k["kernel"]: round(k["total_ms"], 2) for k in
                                            sorted(prof, key=lambda k: -k["total_ms"])[:8]},
                "by_kernel_ms_per_image": {n: round(v, 2) for n, v in sorted(by_kernel.items(), key=lambda kv: -kv[1])}}
        if a.dump_profile:
            with open(a.dump_profile, "w") as f:
                json.dump(sorted(prof, key=lambda k: -k["total_ms"]), f, indent=0)

    if rank == 0:
        images = world * a.steps * B
        per_image_s = elapsed / (a.steps * B)
        flop_img = FLOP_PER_STEP.get(a.res, 0) * a.ddim_steps + FLOP_VAE.get(a.res, 0) if a.config == "sd15" else 0
        out = {
            "metric": "512x512 canny2image images/sec (20 DDIM steps); ms per UNet step reported alongside",
            "value": round(images / elapsed, 4), "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("f16 + block-scaled fp8 x fp8 GEMMs (fp8 e4m3fn weights)" if a.mx else "f16 (fp8 e4m3fn weights)") if a.fp8 else "f16", "data": "synthetic",
            "ms_per_unet_step": round(loop_ms / a.ddim_steps, 3),
            "mfma_frac_whole_image": round(flop_img / per_image_s / 1e12 / PEAK_TFLOPS_F16, 4) if flop_img else None,
            "config": {"workload": f"SD1.5 + ControlNet-canny {a.res}x{a.res}, batch={B} per GPU (CFG pair fused, N={2 * B}), "
                                   f"{a.ddim_steps} DDIM steps + VAE decode, fp16 storage / fp32 accumulate"
                                   + (", UNet / ControlNet weights fp8 e4m3fn with per-output-channel scales (BASELINE configs[4] precision"
                                      + (f"; GEMMs of >= {a.mx_min_rows} rows block-scaled fp8 x fp8 on the fp8 MFMA" if a.mx else "; fp16 MFMA") + ")" if a.fp8
                                      else " (BASELINE configs[1])"),
                       "model_config": a.config, "images_per_gpu_per_step": B, "ddim_steps": a.ddim_steps,
                       "guidance_scale": a.scale, "parallelism": f"dp{world} (image index -> rank, RCCL all_gather of final latents)",
                       "weights": "seeded synthetic (no checkpoint in the container)"},
        }
        if gather_ms is not None:
            out["latent_all_gather_ms"] = round(gather_ms, 3)      # the only collective of the job (rank 0's wall time, sync to sync)
            out["per_rank_images_per_s"] = per_rank
        if golden_check:
            out["output_check"] = golden_check
        if fp8_check:
            out["fp8_check"] = fp8_check
        if roof:
            out["roofline"] = roof
        if world == 1 and not a.no_cpu_baseline and a.config == "sd15":
            out["cpu_baseline"] = cpu_baseline(a.res, a.ddim_steps, a.scale)
        print(json.dumps(out), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
