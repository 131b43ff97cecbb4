#!/usr/bin/env python
"""Headline benchmark: diffusion iter/s (UNet steps/s) of SD2.1-base 512x512 fp16 on MI355X.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one iteration of the reference's denoising loop (pipeline.py:500-573; iter/s as
defined by swift/StableDiffusionCLI/main.swift:225-257): duplicate latents for classifier-free
guidance, one UNet forward at CFG batch 2 per prompt, guidance combine, scheduler (DDIM) update -
all device-resident, inputs already in HBM when the timed region starts.  Weights are random-init
tensors of the SD2.1-base architecture (865.9 M parameters; no checkpoints exist offline) and the
latents / text embeddings are synthetic N(0,1) of the real shapes.

Multi-GPU: independent prompts shard over ranks (one process per GPU, no data-path collective;
RCCL only broadcasts the text embeddings at start and gathers the final latents at the end, both
outside the timed region) -> weak scaling; value = all prompts' steps / max-over-ranks time.

Timing: the K-step timed region (barrier + synchronize on both sides, max over ranks) is repeated
--repeats times (default 5, README.md:79-91 of the reference reports medians); `value` and
`ms_per_step` come from the MEDIAN repeat, every repeat is listed under "repeats_ms_per_step".

Extra objects on the JSON line:
  roofline     MFMA roofline of the step graph: algorithmic FLOP per step (SURVEY.md section 8d:
               1.6085e12 per CFG-batch-2 step) / HIP-event time per step on the handle's stream;
               `traffic` = HBM bytes per step from rocprofv3 PMC passes (profiles/r06_final_hbm_traffic.json,
               `tools/gpu_session.sh <tag> pmc`; only when the file was measured on this very library and workload),
               `dominant_kernel` = the kernel family with the largest TIME share of the step (sd_unet_profile: HIP
               events around every op of the eager step), `kernel_families` = all of them, `flop_heaviest_kernel` =
               the largest conv family in sequence next to its stand-alone time
  --model / --latent select the UNets of BASELINE configs 4 and 5 and 768x768 latents (reported lines; the default
               invocation is BASELINE config 2)
  cpu_baseline the oracle (CPU restatement of the reference UNet + loop) timed on this box's host
               cores on a bounded sample (rank 0, N=1 only: 5 timed steps) at the best of a thread sweep (8 .. 128); kind "port"
  e2e          prompt -> image latency: tokenizer + CLIP text tower on HIP + 20 steps + VAE decode
  gpu_state    rocm-smi clocks / power / temperature before the warm-up and right after the timed repeats
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "ml-stable-diffusion_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

FLOP_PER_SAMPLE_STEP = 1.6085e12 / 2     # 804.3 GFLOP per latent sample (SD2.1-base, 64x64 latents)
MFMA_PEAK_TFLOPS = 2500.0                # MI355X dense fp16/bf16 (MI355X_MICROARCH.md)
PUBLISHED_BEST_ITS = 3.07                # best published SD2.1-base it/s (iPad Pro M2, Core ML)
MODEL = "stabilityai/stable-diffusion-2-1-base"
HBM_TRAFFIC_FILE = "r06_final_hbm_traffic.json"   # written by `tools/gpu_session.sh <tag> pmc` on the final library of the round
# --model: the UNets of BASELINE.json's configs (random-init weights of the real architectures).  The default
# invocation stays BASELINE config 2 (sd21-base at 64x64 latents); the others are reported lines, never `vs_baseline`.
MODELS = {
    "sd21-base": dict(id=MODEL, latent=64, config="BASELINE config 2", name="SD2.1-base (865.9 M-parameter UNet)"),
    "sdxl-base": dict(id="stabilityai/stable-diffusion-xl-base-1.0", latent=96, config="BASELINE config 4 (base stage)",
                      name="SDXL-base-1.0 (2.57 B-parameter UNet, dual-text-encoder conditioning)"),
    "sdxl-refiner": dict(id="stabilityai/stable-diffusion-xl-refiner-1.0", latent=96, config="BASELINE config 4 (refiner stage)",
                         name="SDXL-refiner-1.0 (2.26 B-parameter UNet)"),
    "sd15-control": dict(id="runwayml/stable-diffusion-v1-5", latent=64, config="BASELINE config 5", control=True,
                         name="SD1.5 control-UNet + ControlNet, residuals handed over on the device"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--prompts-per-gpu", type=int, default=1)
    ap.add_argument("--attention", default="ORIGINAL", choices=["ORIGINAL", "SPLIT_EINSUM", "SPLIT_EINSUM_V2"])
    ap.add_argument("--guidance-scale", type=float, default=7.5)
    ap.add_argument("--cpu-steps", type=int, default=5, help="oracle steps timed for cpu_baseline (0 = skip)")
    ap.add_argument("--sustain-s", type=float, default=12.0,
                    help="untimed back-to-back replays of the step graph AFTER the timed repeats (result discarded, never part of "
                         "`value`): long enough for an outside GPU-utilisation sampler to see the run (0 = skip)")
    ap.add_argument("--repeats", type=int, default=5, help="repeats of the K-step timed region (median reported)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--model", default="sd21-base", choices=sorted(MODELS), help="UNet of BASELINE configs 2 / 4 / 5")
    ap.add_argument("--latent", type=int, default=0, choices=[0, 64, 96, 128],
                    help="latent height = width (64: 512x512 images, 96: 768x768); 0 = the model's BASELINE size")
    ap.add_argument("--streams", type=int, default=0,
                    help="extra measurement (N=1 only): S prompts served by S handles on S HIP streams at the same time; reported "
                         "as 'concurrent_prompts', never as 'value'")
    ap.add_argument("--stub-model", action="store_true", help=argparse.SUPPRESS)   # tests/test_parallel.py: the launch /
    # rendezvous / barrier / max-over-ranks / gather plumbing of this script under gloo on CPU, with StubModel in the UNet's place
    args = ap.parse_args()
    spec = MODELS[args.model]
    lat_hw = args.latent or spec["latent"]
    default_cfg = args.model == "sd21-base" and lat_hw == 64

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    stub = args.stub_model
    if not stub and not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if not stub:
        torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if stub:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    sync = (lambda: None) if stub else torch.cuda.synchronize
    red_dev = "cpu" if stub else "cuda"

    from python_hip_stable_diffusion import checkpoint, schedulers
    from python_hip_stable_diffusion.parallel import broadcast_array, gather_arrays, shard_prompts
    if stub:
        HipModel = StubModel
    else:
        from python_hip_stable_diffusion import HipModel

    ppg = args.prompts_per_gpu
    t_build = time.time()
    from python_hip_stable_diffusion.hip_model import UNET_CONFIGS, normalize_unet_config
    ucfg = dict(UNET_CONFIGS[spec["id"]])
    control = None
    if spec.get("control"):
        ucfg["support_controlnet"] = True
    ucfg = normalize_unet_config(ucfg)
    shapes = checkpoint.unet_param_shapes(ucfg)
    ckpt = None if stub else checkpoint.random_checkpoint(shapes, seed=0)
    model = HipModel(ucfg, ckpt, batch=2 * ppg, latent_height=lat_hw, latent_width=lat_hw,
                     attention_implementation=args.attention, device=local_rank, use_graph=not args.no_graph)
    if spec.get("control"):   # pipeline.py:259-284, :519-529: the ControlNet runs inside the UNet handle's step graph
        ccfg = normalize_unet_config(UNET_CONFIGS[spec["id"]])
        control = HipModel(ccfg, checkpoint.random_checkpoint(checkpoint.controlnet_param_shapes(ccfg), seed=2), kind="controlnet",
                           batch=2 * ppg, latent_height=lat_hw, latent_width=lat_hw, attention_implementation=args.attention,
                           device=local_rank, use_graph=not args.no_graph)
        control_cond = np.random.RandomState(95).rand(2 * ppg, 3, lat_hw * 8, lat_hw * 8).astype(np.float16)
        control.set_controlnet_cond(control_cond)
        model.attach_controlnets([control])
    if not default_cfg:
        del ckpt
        ckpt = None
    build_s = time.time() - t_build
    ctx_dim = ucfg["cross_attention_dim"]

    # prompts are independent units: rank r owns global prompts shard_prompts(...)[r]
    mine = shard_prompts(world * ppg, world)[rank]
    ehs = None
    if rank == 0:   # "text embeddings" for all prompts, [uncond | cond] halves per prompt
        ehs = np.random.RandomState(94).randn(world * ppg, 2, ctx_dim, 1, 77).astype(np.float16)
    ehs = broadcast_array(ehs, (world * ppg, 2, ctx_dim, 1, 77), np.float16, dist, local_rank)
    my_ehs = np.concatenate([ehs[mine, 0], ehs[mine, 1]])          # batch order [uncond..., cond...] (pipeline.py:245)
    latents = np.stack([np.random.RandomState(93 + g).randn(4, lat_hw, lat_hw) for g in mine]).astype(np.float32)
    loop_inputs = {"encoder_hidden_states": my_ehs}
    if model.num_time_ids:   # SDXL micro-conditioning (pipeline.py:245-257, StableDiffusionXLPipeline.swift:326-358)
        img = lat_hw * 8
        ids = [img, img, 0, 0, img, img] if model.num_time_ids == 6 else [img, img, 0, 0, 6.0]
        loop_inputs["time_ids"] = np.tile(np.asarray(ids, np.float16), (2 * ppg, 1))
        loop_inputs["text_embeds"] = np.random.RandomState(96).randn(2 * ppg, model.text_embed_dim).astype(np.float16)

    sch = schedulers.DDIMScheduler()

    def run(n_steps):
        sch.set_timesteps(max(n_steps, 1))
        ts, coef, hist = sch.device_tables()
        return model.denoise_loop(latents * sch.init_noise_sigma, ts, coef, args.guidance_scale, history=hist, **loop_inputs)

    def barrier():
        if dist is not None:
            dist.barrier()
        sync()

    gpu_before = gpu_state(local_rank) if rank == 0 and not stub else None
    calib = None
    if rank == 0 and not stub:   # what THIS box can do, measured before the warm-up by four fixed micro-kernels (calib.hip)
        from python_hip_stable_diffusion import _lib as _sdlib
        calib = _sdlib.calibrate(local_rank)
    if args.warmup > 0:
        run(args.warmup)
    rep_s, rep_ev = [], []
    for _ in range(max(1, args.repeats)):
        barrier()
        t0 = time.perf_counter()
        final, ev_ms = run(args.steps)
        sync()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], device=red_dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
            dist.barrier()
        rep_s.append(el)
        rep_ev.append(float(np.median(ev_ms)))
    gpu_after = gpu_state(local_rank) if rank == 0 and not stub else None   # right behind the timed repeats: clocks under load
    sustain = None
    if not stub and args.sustain_s > 0:   # the timed region is 0.5 s of a 60-s run; make the GPU leg visible
        t0 = time.perf_counter()
        n_sus = 0
        while time.perf_counter() - t0 < args.sustain_s:
            run(500)
            n_sus += 500
        sync()
        sustain = {"seconds": round(time.perf_counter() - t0, 2), "steps": n_sus,
                   "note": "untimed back-to-back step-graph replays after the timed repeats; result discarded, not part of `value`"}
    order = np.argsort(rep_s)
    mid = int(order[len(order) // 2])
    elapsed = rep_s[mid]
    assert np.isfinite(final).all()
    all_final = gather_arrays(final, dist, local_rank)               # outside the timed region

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    total_prompt_steps = world * ppg * args.steps
    value = total_prompt_steps / elapsed
    ev_ms_step = rep_ev[mid]
    x = np.concatenate([latents, latents]).astype(np.float16)
    fwd_inputs = dict(loop_inputs, sample=x, timestep=np.full((2 * ppg,), 951, np.float16))
    ops = None
    if stub:         # plumbing test: the line's bookkeeping fields only
        print(json.dumps({"metric": "stub", "value": round(value, 3), "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(elapsed / args.steps * 1e3, 4), "scaling": "weak",
                          "gathered_latents": list(all_final.shape), "prompts": world * ppg,
                          "checksum": float(np.asarray(all_final, np.float64).sum())}), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return
    if world == 1:   # per-op HIP-event times of one eager step (sd_unet_profile), attached ControlNet included
        model(**fwd_inputs)
        ops = model.profile(iters=7)
        if control is not None:   # the ControlNet runs inside the UNet handle's step; profiled as a handle of its own
            control(sample=x, timestep=fwd_inputs["timestep"], encoder_hidden_states=my_ehs, controlnet_cond=control_cond)
            ops = control.profile(iters=7) + ops
    if default_cfg:
        flop_per_launch = FLOP_PER_SAMPLE_STEP * 2 * ppg
        flop_note = "algorithmic: 804.3 GFLOP per latent sample (SURVEY.md section 8d)"
    else:   # executed MFMA FLOP of the launch list (the prompt's K/V projections are hoisted out of the step)
        assert ops is not None, "--model / --latent other than the default are single-GPU lines"
        flop_per_launch = float(sum(fl for _, fl, _ in ops))
        flop_note = "sum of the algorithmic FLOP of every conv / GEMM / attention launch of the step (sd_unet_profile labels)"
    achieved = flop_per_launch / (ev_ms_step * 1e-3) / 1e12
    px = lat_hw * 8
    metric = ("diffusion iter/s (UNet steps/s), SD2.1-base 512x512 fp16" if default_cfg else
              f"diffusion iter/s (UNet steps/s), {args.model} {px}x{px} fp16")
    out = {
        "metric": metric,
        "value": round(value, 3), "unit": "it/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "repeats": len(rep_s), "repeats_ms_per_step": [round(r / args.steps * 1e3, 4) for r in rep_s],
        "vs_baseline": round(value / PUBLISHED_BEST_ITS, 2) if default_cfg and ppg == 1 else None,
        "vs_baseline_note": "BASELINE.md best published SD2.1-base number: 3.07 it/s, iPad Pro (M2), Core ML "
                            "6-bit palettized (README.md:74); no published CPU/GPU-server number exists",
        "dtype": "fp16", "data": "synthetic",
        "config": {"workload": f"{spec['config']}: {spec['name']}, random-init, denoising iteration at {px}x{px} "
                               f"({lat_hw}x{lat_hw} latents), CFG batch 2 per prompt, DDIM, device-resident loop",
                   "unet": args.model, "latent": lat_hw,
                   "attention": args.attention, "prompts_per_gpu": ppg, "global_batch": 2 * ppg * world,
                   "guidance_scale": args.guidance_scale, "hip_graph": not args.no_graph,
                   "parallelism": f"dp{world} (independent prompts per rank, no data-path collective)"},
        "roofline": {"bound": "mfma", "kernel": "unet_step_graph (all MFMA conv/GEMM/attention launches of one step)",
                     "achieved": round(achieved, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved / MFMA_PEAK_TFLOPS, 4), "traffic": None,
                     "flop_per_launch": flop_per_launch, "flop_definition": flop_note, "launch_ms": round(ev_ms_step, 4),
                     "timing": "hipEvent on the handle's stream: median step of the median repeat"},
        "gpu_state": {"before_warmup": gpu_before, "after_timed_region": gpu_after,
                      "note": "rocm-smi clocks / power / junction temperature of this rank's GPU: attributes box-to-box spread"},
        "e2e": {"model_build_s": round(build_s, 1), "final_latents_finite": True,
                "gathered_latents": list(all_final.shape), "hbm_bytes": int(model.device_bytes)},
        "sustained_leg": sustain,
    }
    if calib is not None:
        out.update(calibration_record(calib))
    if world == 1:
        out["roofline"].update(hbm_traffic(ev_ms_step, args, lat_hw))
        out["roofline"].update(kernel_families(ops, ev_ms_step))
    if world == 1 and default_cfg:   # the most FLOP-heavy kernel family of the step: in sequence and alone
        out["roofline"]["flop_heaviest_kernel"] = flop_heaviest_kernel(ops, 2 * ppg)
    if world == 1 and default_cfg:   # end-to-end latency of one generation: 20 DDIM steps + VAE decode (pipeline.py:500-589)
        out["e2e"].update(e2e_latency(model, checkpoint, my_ehs[[0, ppg]], latents[:1], args.guidance_scale,
                                      local_rank))
    if world == 1 and args.streams > 1 and control is None:
        out["concurrent_prompts"] = concurrent_prompts(model, HipModel, ucfg, checkpoint, args, lat_hw, local_rank, loop_inputs, latents)
    if world == 1 and default_cfg and args.cpu_steps > 0:
        out["cpu_baseline"] = cpu_baseline(ckpt, my_ehs[[0, ppg]], latents[:1], args.cpu_steps, args.guidance_scale)
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


# Box calibration (csrc/calib.hip): nine fixed micro-measurements taken before the warm-up.  Eight of them read the same on every
# box of the pool; cold_code_us - what a launch costs more when its CODE is not in the instruction caches - is 0.8 us on the fast
# boxes and 11 us on the slow ones (LAB_NOTES.md Finding 14) and explains the pool's 20-25 % spread of one build.  Round 5 also
# printed a `value_normalised` from a slope fitted per build; VERDICT r5 / ADVICE r5: the slope belongs to ONE build and ONE workload,
# so it is gone - the raw figures stay, next to the reference values of a fast box, and `value` is always the raw measurement.
CALIB_REF = {"copy_gbs": 4750.0, "mfma_tflops": 2030.0, "empty_launch_us": 1.55, "chain_us": 3.62, "handover_us": 6.52,
             "latency_hbm_ns": 360.0, "latency_cache_ns": 72.0, "small_grid_us": 3.04, "cold_code_us": 0.76}


def calibration_record(calib):
    ratios = {k: round((calib[k] / CALIB_REF[k]) if k.endswith(("_us", "_ns")) else (CALIB_REF[k] / calib[k]), 4)
              for k in CALIB_REF if calib.get(k)}
    return {"calibration": dict(calib, reference=CALIB_REF, slowdown_vs_reference=ratios,
                                note="calib.hip, measured before the warm-up: 1-GiB copy GB/s, dense MFMA loop TFLOP/s, us per launch of a "
                                     "323-launch empty graph / of a chain of short kernels on cold operands / of a chain handing 8 MB over "
                                     "between the XCDs' L2s / of a chain of 64-workgroup launches, ns per dependent load from HBM / from the "
                                     "caches, and cold_code_us = us per launch a chain of 32 DIFFERENT 30-KB kernels costs more than the same "
                                     "chain repeating one of them (0.8 on the pool's fast boxes, 11 on its slow ones); slowdown_vs_reference "
                                     "> 1 = this box is slower than the reference box there.  Raw data only: no figure of this line is "
                                     "corrected with it")}


def concurrent_prompts(model, HipModel, ucfg, checkpoint, args, lat_hw, device, loop_inputs, latents):
    """S independent prompts on S handles (one HIP stream + one step graph each, weights per handle) looping at the same time
    (python_hip_stable_diffusion.parallel.run_concurrent) against the same S prompts one after the other on one handle.  The
    headline `value` stays the one-prompt loop; this is the serving-throughput view of the same path on one GPU."""
    from python_hip_stable_diffusion import schedulers
    from python_hip_stable_diffusion.parallel import run_concurrent
    ppg = args.prompts_per_gpu
    handles = [model]
    for i in range(1, args.streams):
        ck = checkpoint.random_checkpoint(checkpoint.unet_param_shapes(ucfg), seed=0)
        handles.append(HipModel(ucfg, ck, batch=2 * ppg, latent_height=lat_hw, latent_width=lat_hw,
                                attention_implementation=args.attention, device=device, use_graph=not args.no_graph))
        del ck
    sch = schedulers.DDIMScheduler()
    sch.set_timesteps(args.steps)
    ts, coef, hist = sch.device_tables()
    jobs = []
    for i, h in enumerate(handles):
        inp = {k: np.roll(v, i, axis=-1) if k == "encoder_hidden_states" else v for k, v in loop_inputs.items()}
        lat = np.roll(latents, i, axis=-1) * sch.init_noise_sigma
        jobs.append(lambda h=h, inp=inp, lat=lat: h.denoise_loop(lat, ts, coef, args.guidance_scale, history=hist, **inp)[0])
    for j in jobs:                             # warm one after the other: every handle captures its graph undisturbed
        j()
    run_concurrent(jobs)
    serial, conc = [], []
    for _ in range(max(1, args.repeats)):
        t0 = time.perf_counter()
        for j in jobs:
            j()
        serial.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        outs = run_concurrent(jobs)
        conc.append(time.perf_counter() - t0)
    assert all(np.isfinite(o).all() for o in outs)
    s_med, c_med = float(np.median(serial)), float(np.median(conc))
    n = args.streams * ppg * args.steps
    return {"streams": args.streams, "prompts": args.streams * ppg,
            "one_after_the_other_it_s": round(n / s_med, 2), "concurrent_it_s": round(n / c_med, 2),
            "ms_per_prompt_step_serial": round(s_med / n * 1e3, 4), "ms_per_prompt_step_concurrent": round(c_med / n * 1e3, 4),
            "note": "host wall clock around whole loops (launch + replay + read-back), S handles = S HIP streams on one GPU"}


class StubModel:
    """Stands in for HipModel under --stub-model (CPU plumbing test of the multi-GPU launch path): the same constructor and
    denoise_loop signature, a deterministic elementwise "UNet" in numpy.  Never used by a real run."""
    num_time_ids = 0
    device_bytes = 0

    def __init__(self, cfg, ckpt, batch=2, latent_height=64, latent_width=64, **kw):
        self.batch = batch

    def denoise_loop(self, latents, ts, coef, guidance, history=0, encoder_hidden_states=None, **kw):
        x = np.asarray(latents, np.float32).copy()
        e = np.asarray(encoder_hidden_states, np.float32)
        n = x.shape[0]                                   # prompt i owns rows i (uncond) and n + i (cond) of the CFG batch
        bias = (e[:n].reshape(n, -1).mean(1) + e[n:].reshape(n, -1).mean(1)).reshape(n, 1, 1, 1)
        for _ in range(len(ts)):
            x = 0.9 * x + 0.01 * bias
        return x, np.full((len(ts),), 0.01, np.float32)


def gpu_state(device=0):
    """sclk / mclk / power / temperature of the GPU from rocm-smi (boxes of the pool differ by up to 25 % on the same
    build - a regression and a slow box must be distinguishable in the record).  Best effort: {} when rocm-smi is not there."""
    import subprocess
    try:
        r = subprocess.run(["rocm-smi", "-d", str(device), "--showclocks", "--showpower", "--showtemp", "--showperflevel", "--json"],
                           capture_output=True, text=True, timeout=20)
        d = json.loads(r.stdout)
        card = d.get(f"card{device}") or next(iter(d.values()))
        keep = {}
        for k, v in card.items():
            kl = k.lower()
            if any(t in kl for t in ("sclk", "mclk", "fclk", "power", "temperature (sensor junction)", "performance level")):
                keep[k] = v
        return keep
    except Exception as e:   # noqa: BLE001 - diagnostics only
        return {"error": f"{type(e).__name__}: {e}"[:120]}


def build_id():
    """Identity of the running build for the committed PMC file: hash of the shipped library (the GPU box has no .git)."""
    import hashlib
    lib = os.path.join(ROOT, "ml-stable-diffusion_amd", "lib", "libsdmi355.so")
    with open(lib, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def hbm_traffic(step_ms, args, lat_hw):
    """HBM bytes per step from committed rocprofv3 PMC passes (bench.py cannot run under the profiler itself):
    profiles/r06_final_hbm_traffic.json is written by tools/pmc_reduce.py from separate --pmc FETCH_SIZE / WRITE_SIZE
    runs of the same step, corrected as MI355X_MICROARCH.md prescribes, and stamped with the hash of the library it
    profiled and the workload.  The number is only emitted when both match this run; otherwise `traffic` is null."""
    path = os.path.join(ROOT, "profiles", HBM_TRAFFIC_FILE)
    if not os.path.exists(path):
        return {"traffic": None, "traffic_source": None, "traffic_detail": {"note": "no PMC file committed for this round"}}
    with open(path) as f:
        t = json.load(f)
    want = {"build_id": build_id(), "model": args.model, "latent": lat_hw, "prompts_per_gpu": args.prompts_per_gpu,
            "attention": args.attention}
    have = {k: t.get(k) for k in want}
    if have != want:
        return {"traffic": None, "traffic_source": None, "traffic_detail": {
            "note": f"{os.path.basename(path)} was measured on another build / workload: not reported as this run's",
            "file": have, "this_run": want}}
    total = float(t["bytes_per_step"])
    return {"traffic": total, "traffic_source": f"replayed from {os.path.relpath(path, ROOT)}: rocprofv3 --pmc passes of the same step on this "
                                                "very library (hash-matched); NOT measured by this run",
            "traffic_detail": {
        "source": f"profiles/{HBM_TRAFFIC_FILE} (rocprofv3 --pmc, eager launches of the same step, same library hash)",
        "build_id": want["build_id"],
        "read_bytes": t.get("read_bytes_per_step"), "write_bytes": t.get("write_bytes_per_step"),
        "algorithmic_min_bytes": t.get("algorithmic_min_bytes"),
        "hbm_gb_per_s_at_this_run": round(total / (step_ms * 1e-3) / 1e9, 1), "hbm_peak_gb_per_s": 8000.0,
        "hbm_frac": round(total / (step_ms * 1e-3) / 8e12, 4)}}


def op_family(label):
    """Kernel family of a launch-list entry (sd_unet_profile label) = the source kernel that runs it."""
    if label.startswith("conv3x3") and "small-N" not in label and " 4->" not in label:
        return "conv3x3 (conv3x3_halo_ks_kernel / stride-2 igemm_kernel)"
    if label.startswith(("gemm1x1", "geglu1x1")):
        return "1x1 GEMMs (igemm_kernel)"
    if label.startswith("attention"):
        return "self-attention (attn8_kernel / attn_kernel)"
    if label.startswith("xattn"):
        return "cross-attention with fused q projection (xattn_kernel)"
    if label.startswith("groupnorm") and " || gemm1x1" in label:
        return "GroupNorm + the resnet's shortcut GEMM in one launch (gn_*_side_kernel)"
    if label.startswith("groupnorm"):
        return "GroupNorm (groupnorm_* kernels)"
    return "other (boundary, time embedding, conv_in / conv_out, residual adds)"


def kernel_families(ops, graph_ms):
    """Time share of every kernel family of the step from the per-op HIP-event medians of one eager step
    (sd_unet_profile), largest first; `dominant_kernel` is the family with the largest TIME share, with its
    fraction of the MFMA roof over all of its launches."""
    fam = {}
    for lbl, fl, ms in ops:
        f = fam.setdefault(op_family(lbl), [0, 0.0, 0.0])
        f[0] += 1
        f[1] += ms
        f[2] += fl
    total = sum(f[1] for f in fam.values())
    rows = []
    for k, (n, ms, fl) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        tf = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        rows.append({"family": k, "launches": n, "ms": round(ms, 4), "share": round(ms / total, 4), "gflop": round(fl / 1e9, 1),
                     "achieved": round(tf, 1), "frac": round(tf / MFMA_PEAK_TFLOPS, 4)})
    top = dict(rows[0])
    top.update({"kernel": top.pop("family"), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "bound": "mfma",
                "timing": "sum over the family's launches of the per-op HIP-event median of one eager step (sd_unet_profile); "
                          "shares are of the eager per-op sum"})
    return {"dominant_kernel": top, "kernel_families": rows, "step_ops": len(ops), "step_ops_ms_sum": round(total, 4),
            "graph_ms": round(graph_ms, 4)}


def flop_heaviest_kernel(ops, B):
    """3x3 convolutions are 49.8 % of the step's FLOPs; the 320->320 conv at 64x64
    (seven per step, 15.1 GFLOP each at CFG batch 2) is the largest family.  `frac` is its IN-SEQUENCE
    time (HIP events around every op of one eager UNet forward, predecessors' caches cold exactly as in the
    graph); the stand-alone time of the same kernel (50 back-to-back launches, operands L2-warm) is reported
    beside it and is NOT the roofline number."""
    from python_hip_stable_diffusion import _lib
    fam = [(lbl, fl, ms) for lbl, fl, ms in ops if lbl.startswith("conv3x3 320->320 @64x64")]
    assert len(fam) == 7, [o[0] for o in ops if o[0].startswith("conv3x3")][:12]
    ms_in = float(np.mean([m for _, _, m in fam]))
    flop = fam[0][1]
    tf_in = flop / (ms_in * 1e-3) / 1e12
    rs = np.random.RandomState(7)
    xs = rs.randn(B, 320, 64, 64).astype(np.float16)
    w = (rs.randn(320, 320, 3, 3) / np.sqrt(320 * 9)).astype(np.float16)
    _, ms = _lib.conv2d(xs, w, np.zeros(320, np.float32), None, iters=50)
    tf = flop / (ms * 1e-3) / 1e12
    return {"kernel": f"3x3 conv 320->320 @64x64, UNet batch {B} (K-split software-pipelined LDS-halo MFMA kernel, plan from "
                      "tuned_convs.inc), 7 launches per step",
            "flop_per_launch": flop, "launch_ms": round(ms_in, 5), "achieved": round(tf_in, 1), "peak": MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": round(tf_in / MFMA_PEAK_TFLOPS, 4),
            "timing": "in sequence: mean over the family's 7 launches of the per-op HIP-event median (sd_unet_profile)",
            "standalone": {"launch_ms": round(ms, 5), "achieved": round(tf, 1), "frac": round(tf / MFMA_PEAK_TFLOPS, 4),
                           "timing": "same kernel alone, 50 back-to-back launches, operands L2-warm"}}


def synthetic_clip_tokenizer(tmp_dir):
    """transformers' CLIPTokenizer over a byte-level vocabulary without merges (no tokenizer files exist offline): the same
    code path, padding and truncation to 77 tokens as pipeline.py:146-150; only the vocabulary is synthetic."""
    from transformers import CLIPTokenizer
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAD)) + list(range(0xAE, 0x100))
    cs, n = bs[:], 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    alpha = [chr(c) for c in cs]
    vocab = {}
    for ch in alpha:
        vocab[ch] = len(vocab)
    for ch in alpha:
        vocab[ch + "</w>"] = len(vocab)
    vocab["<|startoftext|>"] = len(vocab)
    vocab["<|endoftext|>"] = len(vocab)
    os.makedirs(tmp_dir, exist_ok=True)
    with open(os.path.join(tmp_dir, "vocab.json"), "w") as f:
        json.dump(vocab, f)
    with open(os.path.join(tmp_dir, "merges.txt"), "w") as f:
        f.write("#version: 0.2\n")
    return CLIPTokenizer(os.path.join(tmp_dir, "vocab.json"), os.path.join(tmp_dir, "merges.txt"), model_max_length=77)


def e2e_latency(model, checkpoint, ehs, latents, guidance, device):
    """Median of 3 back-to-back generations (README.md:79-91 methodology), PROMPT -> IMAGE like pipeline.py:123-320: tokenise
    the prompt and the empty negative prompt, the CLIP text tower on the HIP kernels for both (random-init weights of SD2.1's
    OpenCLIP-H architecture), 20 DDIM steps device-resident, fp16 VAE decode.  The denoising loop runs on the bench's fixed
    synthetic embeddings of the same shape (random-init towers produce embeddings of no meaning either way; the timing is what
    is reported), so `latency_20_steps_plus_vae_s` (round 3's field) stays comparable."""
    import tempfile

    from python_hip_stable_diffusion import HipVaeDecoder, VAE_CONFIGS, schedulers
    from python_hip_stable_diffusion.text_encoder import HipTextEncoder
    if model.batch != 2:
        return {}
    vcfg = VAE_CONFIGS[MODEL]
    vae = HipVaeDecoder(vcfg, checkpoint.random_checkpoint(checkpoint.vae_decoder_param_shapes(vcfg), seed=1),
                        batch=1, latent_height=64, latent_width=64, device=device)
    tcfg = checkpoint.TEXT_ENCODER_CONFIGS[MODEL]
    enc = HipTextEncoder(tcfg, checkpoint.random_checkpoint(checkpoint.text_encoder_param_shapes(tcfg), seed=3), device=device)
    tok = synthetic_clip_tokenizer(os.path.join(tempfile.gettempdir(), "sd_bench_tokenizer"))
    prompt = "a high quality photo of an astronaut riding a horse in space"
    sch = schedulers.DDIMScheduler()
    sch.set_timesteps(20)
    ts, coef, hist = sch.device_tables()
    times, vae_ms, text_ms, total = [], [], [], []
    for i in range(4):
        t0 = time.perf_counter()
        emb = []
        for text in ("", prompt):                                    # pipeline.py:151-175, :201-243: negative + positive prompt
            ids = tok(text, padding="max_length", max_length=77, truncation=True, return_tensors="np").input_ids
            emb.append(enc(input_ids=ids.astype(np.float32))["last_hidden_state"])
        t_text = time.perf_counter()
        lat, _ = model.denoise_loop(latents, ts, coef, guidance, history=hist, encoder_hidden_states=ehs)
        t1 = time.perf_counter()
        img = vae(z=(lat / 0.18215).astype(np.float16))["image"]
        t2 = time.perf_counter()
        if i:
            times.append(t2 - t_text)
            vae_ms.append((t2 - t1) * 1e3)
            text_ms.append((t_text - t0) * 1e3)
            total.append(t2 - t0)
    assert np.isfinite(img).all() and all(np.isfinite(e).all() for e in emb) and emb[0].shape == (1, 77, tcfg["hidden_size"])
    vae.close()
    enc.close()
    return {"latency_prompt_to_image_s": round(float(np.median(total)), 4),
            "latency_20_steps_plus_vae_s": round(float(np.median(times)), 4), "vae_decode_ms": round(float(np.median(vae_ms)), 2),
            "tokenize_and_text_encoder_ms": round(float(np.median(text_ms)), 2),
            "note": "prompt -> image, host wall clock: CLIPTokenizer (synthetic byte-level vocabulary) + OpenCLIP-H text tower on HIP "
                    "for the prompt and the empty negative prompt, 20 DDIM steps device-resident, fp16 VAE decode to 512x512; "
                    "latency_20_steps_plus_vae_s excludes the text stage (round-3 definition)"}


def cpu_baseline(ckpt, ehs, latents, n_steps, guidance):
    """The oracle (CPU port of the reference UNet, oracle/unet_ref.py, driven by the restated
    pipeline loop, oracle/scheduler_ref.py) on this box's host cores: a small thread-count sweep on single
    UNet forwards (more torch threads than ~1 per physical core oversubscribe: round 1's 128-thread run was
    2.7x slower than 8 threads in the build container), then n timed + 1 warm-up steps of the same workload
    (same weights, same shapes) at the best setting.  Checker code used as a reported baseline, never as the
    product path."""
    import torch

    from oracle import scheduler_ref, unet_ref, weights
    cfg = unet_ref.CONFIGS["sd21-base"]
    sd = weights.to_torch({k: v.astype(np.float32) for k, v in ckpt.items()})
    times = []

    def unet(x, t, e):
        t0 = time.perf_counter()
        y = unet_ref.unet_forward(sd, cfg, torch.from_numpy(x.astype(np.float32)), torch.from_numpy(t.astype(np.float32)),
                                  torch.from_numpy(e.astype(np.float32))).numpy()
        times.append(time.perf_counter() - t0)
        return y

    t_all = time.perf_counter()
    ncpu = os.cpu_count() or 1
    x = np.concatenate([latents, latents]).astype(np.float16)
    ts = np.array([951, 951], np.float16)
    sweep = {}
    for th in [c for c in (8, 16, 32, 64, 128) if c <= ncpu] or [ncpu]:
        torch.set_num_threads(th)
        if not sweep:
            unet(x, ts, ehs)             # one warm-up forward (allocator, oneDNN primitive caches)
        unet(x, ts, ehs)
        sweep[th] = round(times[-1], 3)
        if len(sweep) >= 3 and sweep[th] > 1.5 * min(sweep.values()):
            break                        # oversubscribed: more threads only get slower (and the sweep must stay within ~30 s)
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    times.clear()
    scheduler_ref.denoise_loop(unet, scheduler_ref.DDIM(), latents, ehs, n_steps + 1, guidance)
    total = time.perf_counter() - t_all
    per = float(np.median(times[1:]))
    out = {"value": round(1.0 / per, 4), "unit": "it/s", "cores": int(best), "host_cpus": ncpu, "kind": "port",
           "sample": f"{n_steps} timed + 1 warm-up CFG-batch-2 steps of the same SD2.1-base loop, torch-CPU fp32, at the "
                     f"best of a thread sweep ({total:.1f} s of CPU work in all)",
           "s_per_step": round(per, 3), "thread_sweep_s_per_forward": sweep}
    ref = os.path.join(ROOT, "profiles", "r02_cpu_reference_build_container.json")
    if os.path.exists(ref):              # the reference's own py/unet.py, timed where /root/reference exists
        with open(ref) as f:
            out["reference_in_build_container"] = json.load(f)
    return out


if __name__ == "__main__":
    main()
