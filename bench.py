#!/usr/bin/env python
"""Headline benchmark: audio-seconds/sec for the Whisper-large-v3 -> Llama-3-8B LoRA training step on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W`; W untimed steps, then EXACTLY K steps bracketed by
barrier + synchronize, MAX over ranks, rank 0 prints ONE JSON line.  N > 1: one rank per GPU over RCCL; when the
script is started WITHOUT a torchrun environment it re-launches itself under `python -m torch.distributed.run
--nproc-per-node N` (rendezvous on 127.0.0.1); when fewer than N GPUs are visible the ranks share devices over gloo
(functional check of the N > 1 path on a 1-GPU box; the JSON line says so in `backend`).

A "step" = one full optimizer step of the hot path on one batch of synthetic input already resident in HBM:
GPU log-mel -> Whisper encoder -> projector -> embed/splice -> Llama-3-8B (+LoRA) forward -> CE/accuracy -> backward ->
(N>1: all-reduce of the flat gradient buffer, overlapped with the backward) -> fused AdamW + LR scheduler.

Workloads (BASELINE.json `configs`):
  c3 (default, the headline): configs[2], Whisper-large-v3 -> Llama-3-8B, linear projector, LoRA r16 (q,v), 31 x 30 s clips per
      GPU (31 x T=380 = 11 780 <= max_frame_length 12 000: the reference's window_class admits 31 and refuses the 32nd);
  c1: configs[0], Whisper-tiny -> TinyLlama-1.1B, LoRA r8, 1 x 10 s clip (padded to 30 s by the recipe; plumbing-sized);
  c2: configs[1], Whisper-base -> Llama-3-8B, LoRA r16, batch 8 x 30 s;
  c4: configs[3], HuBERT-large -> Vicuna-7B, Q-Former (32 queries, 8 layers), LoRA r32, batch 6 x 30 s raw waveforms.
Weak scaling: per-GPU work is fixed as N grows; no data-path collective except the gradient all-reduce.
"""
import argparse
import json
import os
# HIP runtime configuration of the product (INTEGRATION.md "Runtime environment"): kernel arguments written straight into device memory instead of a host-visible
# buffer the command processor reads over the bus.  Measured in-step on MI355X, three interleaved rounds (profiles/r06_runtime_env.md): C4 48.75 -> 46.44 ms, C1 13.02 -> 12.42,
# C2 86.34 -> 84.96, C3 350.4 -> 348.2.  Set before the HIP runtime is loaded (i.e. before `import torch`); an exported value wins.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import socket
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # dense MFMA bf16 peak, /opt/skills/guides/MI355X_MICROARCH.md
# Per-launch HIP events (the roofline's live measurement) are recorded on every KERNEL_TIMING_EVERY-th step of the timed region, not on
# all of them: ~1 170 events per C3 step cost the step they measure 3.4 ms = 0.9 % (tools/timer_overhead.py, profiles/r05_timer_overhead.md).
# The sampled steps are ordinary steps of the timed region (step 0, 4, 8, ...: with the default K = 5 two of five, with K = 20 five of twenty).
KERNEL_TIMING_EVERY = int(os.environ.get("SLAM_BENCH_TIMING_EVERY", "4"))
CLIP_SECONDS, PROMPT, ANSWER = 30.0, 16, 64

WORKLOADS = {
    "c1": dict(title="C1", clips=1, clip_seconds=10.0,
               model=dict(encoder_name="whisper", encoder_path="tiny.pt", llm_name="tinyllama-1.1b", encoder_dim=384, llm_dim=2048,
                          encoder_projector="linear", encoder_projector_ds_rate=5),
               peft=dict(r=8, lora_alpha=32, target_modules=["q_proj", "v_proj"], lora_dropout=0.05)),
    "c3": dict(title="C3", clips=31, model=dict(encoder_name="whisper", encoder_path="large-v3.pt", llm_name="llama-3-8b", encoder_dim=1280,
                                                 encoder_projector="linear", encoder_projector_ds_rate=5),
               peft=dict(r=16, lora_alpha=32, target_modules=["q_proj", "v_proj"], lora_dropout=0.05)),
    "c2": dict(title="C2", clips=8, model=dict(encoder_name="whisper", encoder_path="base.pt", llm_name="llama-3-8b", encoder_dim=512,
                                                encoder_projector="linear", encoder_projector_ds_rate=5),
               peft=dict(r=16, lora_alpha=32, target_modules=["q_proj", "v_proj"], lora_dropout=0.05)),
    "c4": dict(title="C4", clips=6, model=dict(encoder_name="hubert", encoder_path="hubert_large_ll60k.pt", llm_name="vicuna-7b-v1.5",
                                                encoder_dim=1024, encoder_projector="q-former", qformer_layers=8, query_len=32),
               peft=dict(r=32, lora_alpha=32, target_modules=["q_proj", "v_proj"], lora_dropout=0.05)),
}


def algorithmic_flops_per_clip(cfg, T, Ta, n_frames):
    """SURVEY.md 8(d) FLOP model (pads and recomputation never count)."""
    d, L, V = cfg["llm_dim"], cfg["llm_layers"], cfg["vocab"]
    dkv = cfg["llm_kv_heads"] * cfg["llm_head_dim"]
    if cfg.get("encoder_name") == "hubert":
        # same counting rules applied to the HuBERT graph: conv stack, feature projection, grouped positional
        # conv, L_e pre-LN layers; Q-Former: self-attn over Q queries every layer, cross-attn (K/V projections over the T_e
        # encoder frames) every 2nd layer, FFN, output Linear -- forward + dX + dW = x3 (trainable)
        de, Le = cfg["hub_dim"], cfg["hub_layers"]
        n, cin, f_conv = 480000, 1, 0
        for co, k, s_ in zip(cfg["hub_conv_dim"], cfg["hub_conv_kernel"], cfg["hub_conv_stride"]):
            n = (n - k) // s_ + 1
            f_conv += 2 * n * k * cin * co
            cin = co
        Te = n
        f_enc = (f_conv + 2 * Te * cin * de + 2 * Te * de * cfg["hub_pos_k"] * (de // cfg["hub_pos_groups"])
                 + Le * (2 * Te * (4 * de * de + 2 * de * cfg["hub_ffn"]) + 4 * Te * Te * de))
        dq, Q, Lq, Fq = cfg["qf_dim"], cfg["qf_queries"], cfg["qf_layers"], cfg["qf_ffn"]
        n_cross = (Lq + cfg["qf_cross_freq"] - 1) // cfg["qf_cross_freq"]
        f_q = Lq * (2 * Q * 4 * dq * dq + 4 * Q * Q * dq + 2 * Q * 2 * dq * Fq)
        f_q += n_cross * (2 * Q * 2 * dq * dq + 2 * Te * 2 * de * dq + 4 * Q * Te * dq)
        f_proj = 3 * (f_q + 2 * Q * dq * d)
        f_mel = 0
    else:
        de, Le, nm = cfg["enc_dim"], cfg["enc_layers"], cfg["n_mels"]
        Te = (n_frames + 1) // 2
        f_enc = Le * (24 * Te * de * de + 4 * Te * Te * de) + 2 * 3 * n_frames * nm * de + 2 * 3 * Te * de * de
        if not cfg.get("freeze_encoder", True):
            f_enc *= 3   # forward + dX + dW
        f_proj = 3 * 2 * Ta * (cfg["ds_rate"] * de * cfg["proj_hidden"] + cfg["proj_hidden"] * d)
        f_mel = n_frames * (2 * 400 * 402 + 2 * 201 * nm)
    p_mm = L * (d * (d + 2 * dkv + d) + 3 * d * cfg["llm_ffn"]) + V * d
    f_llm = 2 * (2 * T * p_mm + L * 2 * T * T * d)
    tgt = {"q_proj": d + d, "k_proj": d + dkv, "v_proj": d + dkv, "o_proj": d + d}
    f_lora = 3 * 2 * T * cfg["lora_r"] * sum(tgt[t] for t in cfg["lora_targets"]) * L
    return dict(enc=f_enc, proj=f_proj, llm=f_llm, lora=f_lora, mel=f_mel, total=f_enc + f_proj + f_llm + f_lora + f_mel)


def audio_tokens(cfg):
    if cfg.get("projector") == "q-former":
        return cfg["qf_queries"]          # fix_length_audio == query_len (SURVEY a3')
    if cfg.get("encoder_name") == "hubert":
        return 480000 // 320 // 5         # speech_dataset.py:98-99
    return ((3000 + 1) // 2) // cfg["ds_rate"]


def make_batch(cfg, n_clips, dev, seed, clip_seconds=CLIP_SECONDS):
    """synthetic batch in the reference's dict layout (speech_dataset_large.py:180-233 collator)."""
    import torch
    g = torch.Generator(device=dev).manual_seed(seed)
    audio = (torch.randn(n_clips, int(clip_seconds * 16000), generator=g, device=dev) * 0.1).clamp_(-1, 1)
    if cfg.get("encoder_name") == "hubert":   # dataset_config.normalize (speech_dataset.py:96-97)
        audio = torch.nn.functional.layer_norm(audio, (audio.shape[1],))
    Ta = audio_tokens(cfg)
    T = Ta + PROMPT + ANSWER
    ids = torch.randint(3, cfg["vocab"], (n_clips, T), generator=g, device=dev, dtype=torch.int64)
    ids[:, :Ta] = -1
    ids[:, -1] = 2  # eos
    labels = ids.clone()
    labels[:, : Ta + PROMPT] = -100
    mm = torch.zeros((n_clips, T), dtype=torch.bool, device=dev)
    mm[:, :Ta] = True
    return dict(input_ids=ids, labels=labels, attention_mask=torch.ones((n_clips, T), dtype=torch.bool, device=dev),
                modality_mask=mm, audio=audio), T, Ta


def physical_cores():
    """(sockets x cores) from /proc/cpuinfo; falls back to os.cpu_count()"""
    try:
        seen = set()
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        return len(seen) or os.cpu_count()
    except OSError:
        return os.cpu_count()


def cpu_baseline(cfg, clip_seconds=CLIP_SECONDS, n_warm=1, n_meas=3):
    """The CPU leg: the oracle -- the fp32 restatement of the reference path, pinned to the reference by
    tests/golden/ (kind "port"; /root/reference itself is not present on the GPU box) -- timed on the host cores on a
    BOUNDED sample of the same workload: ONE 30 s clip at the true layer dimensions, with the encoder / LLM depth reduced
    to (2,2), (4,2) and (2,4) layers; each point = median of 3 optimizer steps after 1 warm-up; the per-layer costs are
    solved from the three points and extrapolated linearly to the full depth."""
    import torch
    from oracle import slam_oracle as O
    hub = cfg.get("encoder_name") == "hubert"
    # pick the host thread count that actually runs torch's CPU GEMM fastest (all hardware threads oversubscribe at M=380 rows)
    a, b = torch.randn(380, 4096), torch.randn(14336, 4096)
    best, threads = None, os.cpu_count()
    for nt in sorted({min(os.cpu_count(), n) for n in (8, 16, 32, 64, 128, os.cpu_count())}):
        torch.set_num_threads(nt)
        torch.nn.functional.linear(a, b)
        t0 = time.perf_counter()
        for _ in range(3):
            torch.nn.functional.linear(a, b)
        dt = time.perf_counter() - t0
        if best is None or dt < best:
            best, threads = dt, nt
    torch.set_num_threads(threads)
    Le_key = "hub_layers" if hub else "enc_layers"
    Le_full, Ll_full = cfg[Le_key], cfg["llm_layers"]

    def one_point(le, ll):
        c = dict(cfg, **{Le_key: le, "llm_layers": ll})
        W = O.init_weights(dict(c, enc_layers=1) if hub else c, seed=42)   # (the HuBERT case replaces the encoder/projector entries below)
        audio = O.synth_audio(1, clip_seconds, seed=1234)
        if hub:
            W = {k: v for k, v in W.items() if not k.startswith(("encoder.", "encoder_projector."))}
            W.update(O.init_hubert_weights(c, seed=7))
            W.update(O.init_qformer_weights(c, c["enc_dim"], c["llm_dim"], seed=11))
            wav = torch.nn.functional.layer_norm(audio, (audio.shape[1],))
            Q = c["qf_queries"]
            g = torch.Generator().manual_seed(1236)
            s = O.make_sample(Q, torch.randint(3, c["vocab"], (PROMPT,), generator=g).tolist(),
                              torch.randint(3, c["vocab"], (ANSWER - 1,), generator=g).tolist(), 2)
            batch = O.collate_right_pad([s], pad_id=2)
            names = O.trainable_names(W)
            for n in W:
                W[n].requires_grad_(n in names)
            opt = torch.optim.AdamW([W[n] for n in names], lr=1e-4, weight_decay=0.0)

            def step():
                enc = O.hubert_encoder(W, c, wav)
                proj = O.projector_qformer(W, c, enc, None)
                emb = O.embed_splice(W["llm.base_model.model.model.embed_tokens.weight"], batch["input_ids"].clone(),
                                     batch["modality_mask"].bool(), proj)
                loss, _ = O.llama_forward(W, c, emb, batch["attention_mask"], batch["labels"])
                loss.backward()
                opt.step()
                opt.zero_grad()
        else:
            batch = O.synth_batch(c, audio, prompt_len=PROMPT, answer_lens=(ANSWER,), seed=1236, left_pad=False)

            def step():
                O.train_steps(W, c, [batch], lr=1e-4)
        ts = []
        for i in range(n_warm + n_meas):
            t0 = time.perf_counter()
            step()
            ts.append(time.perf_counter() - t0)
        return statistics.median(ts[n_warm:]), ts

    t22, r22 = one_point(2, 2)
    t42, r42 = one_point(4, 2)
    t24, r24 = one_point(2, 4)
    b_e, b_l = (t42 - t22) / 2, (t24 - t22) / 2
    fixed = t22 - 2 * b_e - 2 * b_l
    t_full = fixed + Le_full * b_e + Ll_full * b_l
    fmt = lambda r: "[" + ", ".join(f"{x:.2f}" for x in r) + "]"  # noqa: E731
    return dict(value=clip_seconds / t_full, unit="audio-seconds/sec", cores=physical_cores(), threads=threads, host_logical_cpus=os.cpu_count(),
                kind="port", step_seconds_full_depth=t_full,
                raw_step_seconds={"(2,2)": r22, "(4,2)": r42, "(2,4)": r24},
                sample=(f"oracle (CPU restatement of the reference path pinned by tests/golden; the reference itself is absent on the GPU "
                        f"box) fp32 train step on 1 x {clip_seconds:g} s clip (recipe-padded to 30 s) at true layer dims, {threads} torch threads on {physical_cores()} physical cores; "
                        f"median of {n_meas} steps after {n_warm} warm-up at (enc,llm) layers (2,2)={t22:.2f}s {fmt(r22)}, (4,2)={t42:.2f}s "
                        f"{fmt(r42)}, (2,4)={t24:.2f}s {fmt(r24)}; per-layer cost enc {b_e:.3f}s llm {b_l:.3f}s, fixed {fixed:.2f}s -> "
                        f"({Le_full},{Ll_full}) layers = {t_full:.1f} s/clip"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def fail(args, message, rc=2):
    """a multi-GPU run that cannot be what it claims (fewer devices than ranks, a backend other than RCCL, ranks that did not all
    join) must not print a plausible-looking number: rank 0 prints ONE JSON line with value null + the reason, exit code != 0
    (the first hardware run must not be able to fall back to gloo silently)"""
    if int(os.environ.get("RANK", "0")) == 0:
        print(json.dumps({"metric": "audio-seconds/sec/node (Whisper-large-v3->Llama-3-8B LoRA)", "value": None, "unit": "audio-seconds/sec",
                          "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "error": message}), flush=True)
    sys.exit(rc)


def self_launch(args):
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks ourselves."""
    import torch
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if torch.cuda.device_count() < args.gpus:
        if not args.functional_gloo:
            fail(args, f"--gpus {args.gpus} but only {torch.cuda.device_count()} HIP device(s) are visible: a scaling line needs one MI355X per rank "
                       f"over RCCL.  (--functional-gloo runs the N > 1 code path with ranks SHARING devices over gloo: a functional check, "
                       f"marked invalid_for_scaling in its line.)")
        env["SLAM_DIST_BACKEND"] = "gloo"   # ranks share a device: RCCL cannot, gloo can (functional check only)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    os.execvpe(cmd[0], cmd, env)


def traffic_for(kernel_name):
    """HBM-side bytes per launch of the dominant kernel from the committed PMC summary (separate --pmc passes over this
    same command, FETCH_SIZE with the gfx950 x2 correction; tools/pmc_traffic.py writes the file)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(path):
        return None, None
    table = json.load(open(path))
    want = kernel_name.rstrip(">")
    for k, v in table.get("kernels", {}).items():   # (the profiler prints every template argument, ops names the leading ones)
        have = k.rstrip(">")
        if have.startswith(want) or want.startswith(have):
            return v, table.get("source")
    return None, table.get("source")


def pmc_for(kernel_name):
    """MFMA utilisation and effective clock of the dominant kernel from the committed PMC summary (profiles/pmc.json: rocprofv3 --pmc
    SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE ... over tools/pmc_gemm.py; the counters cannot be read inside this process)."""
    path = os.path.join(ROOT, "profiles", "pmc.json")
    if not os.path.exists(path):
        return None
    table = json.load(open(path))
    want = kernel_name.rstrip(">")
    rows = {k: v for k, v in table.get("kernels", {}).items() if k.split(" @ ")[0].rstrip(">").startswith(want) or want.startswith(k.split(" @ ")[0].rstrip(">"))}
    if not rows:
        return None
    return dict(source=table.get("source"), units=table.get("units"),
                per_shape={k.split(" @ ")[-1]: dict(mfma_util=v["mfma_util"], effective_clock_GHz=v["effective_clock_GHz"], duration_us=v["duration_us"],
                                                    wave_cycles_share=v["wave_cycles_share"], l2_hit_rate=v["l2_hit_rate"]) for k, v in rows.items()})


def rank_seed(rank: int) -> int:
    """every rank draws its own synthetic batch (weak scaling: per-GPU work fixed, data different)"""
    return 1234 + rank


def measure(step, steps: int, warmup: int, world: int, dist, sync, device, comm_ms_fn=None):
    """The timed region of the contract, the same on every rank: W untimed steps, sync + barrier, EXACTLY K steps, sync + barrier; the
    elapsed time and the exposed-communication time are the MAX over ranks (one all-reduce of two doubles).  `sync` = the device
    synchronisation (torch.cuda.synchronize; a no-op for the CPU / gloo test of this logic), `comm_ms_fn` = this rank's mean exposed
    communication per step (GradSync) or None.  Returns (last step's result, elapsed seconds, comm_exposed_ms | None)."""
    import time as _time
    import torch
    res = None
    for _ in range(warmup):
        res = step()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    t0 = _time.perf_counter()
    for _ in range(steps):
        res = step()
    sync()
    if world > 1:
        dist.barrier()
    elapsed = _time.perf_counter() - t0
    comm = comm_ms_fn() if comm_ms_fn is not None else None
    if world > 1:
        te = torch.tensor([elapsed, comm or 0.0], dtype=torch.float64, device=device)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te[0].item())
        if comm is not None:
            comm = float(te[1].item())
    return res, elapsed, comm


def throughput(world: int, n_clips: int, clip_s: float, steps: int, elapsed: float):
    """WHOLE-JOB audio-seconds/sec over all ranks and ms per step"""
    return world * n_clips * clip_s * steps / elapsed, elapsed / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--clips", type=int, default=0, help="clips per GPU (default: the workload's)")
    ap.add_argument("--train-encoder", action="store_true", help="train_config.freeze_encoder=false (Whisper workloads): the encoder "
                    "trains too -- NOT the headline configuration, named as such in config.workload")
    ap.add_argument("--functional-gloo", action="store_true", help="N>1 on a box with fewer than N GPUs: ranks share devices over gloo (functional "
                    "check of the N > 1 path only; without this flag such a run FAILS instead of printing a number)")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"], help="replay the training step as ONE captured HIP graph "
                    "(slam_llm_amd.train.GraphedTrainStep: bit-identical to the eager step) on the steps that carry no per-launch timing events; "
                    "auto = off (measured neutral on every workload: profiles/r06_graph_ab.md)")
    ap.add_argument("--ddp", action="store_true", help="N>1: reduce through torch DistributedDataParallel (autograd_params mode) "
                                                      "instead of the GradSync fast path")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)

    import torch
    import torch.distributed as dist
    from slam_llm_amd import ops
    from slam_llm_amd.model import SlamAdamW, SlamHipModel
    from slam_llm_amd.slam_model_hip import build_config
    from slam_llm_amd.train import GradSync, GraphedTrainStep, lr_lambda, rccl_version, setup_distributed, train_step

    rank, local_rank, world = setup_distributed("cuda")
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    backend = dist.get_backend() if world > 1 else None
    n_ranks_seen, devices_distinct = 1, True
    if world > 1:
        # what the collective library itself sees: a SUM of ones over the group (must be N), and every rank's device identity
        # (PCI bus id where the runtime exposes it, else the index) gathered through the same backend
        one = torch.ones(1, device=dev)
        dist.all_reduce(one)
        n_ranks_seen = int(one.item())
        props = torch.cuda.get_device_properties(local_rank)
        ident = f"{getattr(props, 'pci_bus_id', -1)}:{getattr(props, 'pci_device_id', -1)}:{getattr(props, 'uuid', local_rank)}:{local_rank}"
        idents = [None] * world
        dist.all_gather_object(idents, ident)
        devices_distinct = len(set(idents)) == world
        if n_ranks_seen != args.gpus:
            fail(args, f"the process group reduced {n_ranks_seen} ranks, --gpus says {args.gpus}")
        if (backend != "nccl" or not devices_distinct) and not args.functional_gloo:
            fail(args, f"N = {world} ranks over backend '{backend}' with {len(set(idents))} distinct device(s): a scaling line needs backend nccl (= RCCL) "
                       f"and one MI355X per rank (SLAM_DIST_BACKEND / a shared device is only accepted with --functional-gloo)")

    # recipe defaults (examples/asr_librispeech/asr_config.py:29-37): LoRA on q_proj,v_proj with lora_dropout 0.05 live in
    # train mode ("dropout 0 for parity, 0.05 for throughput"); r per BASELINE.json
    wl = WORKLOADS[args.workload]
    n_clips = args.clips or wl["clips"]
    cfg = build_config(dict(use_peft=True, peft_config=wl["peft"], seed=42, freeze_encoder=not args.train_encoder), wl["model"])
    model = SlamHipModel(cfg, dev, autograd_params=bool(args.ddp and world > 1)).init_random(42)
    model.train()
    gsync = None
    step_model = model
    if world > 1 and args.ddp:
        step_model = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank] if backend == "nccl" else None)
    elif world > 1:
        gsync = GradSync(model).attach(model)
    opt = SlamAdamW(model, lr=1e-4, weight_decay=0.0)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda s: lr_lambda(s, 1000, 100000))
    clip_s = wl.get("clip_seconds", CLIP_SECONDS)
    batch, T, Ta = make_batch(cfg, n_clips, dev, seed=rank_seed(rank), clip_seconds=clip_s)

    timer = ops.KernelTimer()
    sampled = {"on": False, "i": 0, "n": 0}
    # auto = off: the captured step measured NEUTRAL on every workload once the per-launch timing events are sampled sparsely (C3 350.0 vs
    # 350.3 ms, C2 85.2 vs 85.1, C4 46.5 vs 46.8, C1 11.48 vs 11.53: profiles/r06_graph_ab.md) -- the host already runs ahead of the
    # device in the eager loop, and a graph's kernels start no closer together than a stream's.  The lines stay on the eager step.
    use_graph = args.graph == "on" and not args.ddp
    # static bound on the labelled rows = what the synthetic collator produces (ANSWER labelled rows per clip); a real loader passes
    # B x its longest answer.  The eager steps (warm-up, the sampled timing steps) run under the same bound.
    graphed = GraphedTrainStep(model, opt, sched, label_rows_cap=n_clips * ANSWER, warmup=1, grad_sync=gsync) if use_graph else None

    def step():
        on = sampled["on"] and sampled["i"] % KERNEL_TIMING_EVERY == 0
        sampled["i"] += 1
        sampled["n"] += 1 if on else 0
        ops.TIMER = timer if on else None       # (per-launch events on the sampled steps only: see KERNEL_TIMING_EVERY)
        try:
            if graphed is not None:             # (falls back to the eager train_step by itself while ops.TIMER is set)
                return graphed(batch)
            return train_step(step_model, batch, opt, sched, gsync)
        finally:
            ops.TIMER = None

    # warm-up outside the kernel timer, then the timed region (measure(): barrier + synchronize on both sides, MAX over ranks)
    measure(step, 0, args.warmup, world, dist, torch.cuda.synchronize, dev)
    sampled.update(on=True, i=0, n=0)
    if gsync is not None:
        gsync.time_finish = True    # HIP events around finish(): the part of the gradient exchange the backward did not hide
    (loss, acc), elapsed, comm_exposed_ms = measure(step, args.steps, 0, world, dist, torch.cuda.synchronize, dev,
                                                    comm_ms_fn=(gsync.exposed_ms_per_step if gsync is not None else None))
    n_timed = max(1, sampled["n"])           # steps of the timed region that carried per-launch events
    if rank != 0:
        if world > 1:
            dist.barrier()
        return

    value, ms_per_step = throughput(world, n_clips, clip_s, args.steps, elapsed)
    fl = algorithmic_flops_per_clip(cfg, T, Ta, 3000)
    step_flops = fl["total"] * n_clips
    # products the step does not execute for rows without a label (slam_llm_amd.model.LM_HEAD_LABEL_ROWS / LAST_LAYER_LABEL_ROWS): the
    # lm_head forward + dX and, in the last decoder layer, o / gate / up / down forward + dX of those rows -- `mfu` counts what ran
    from slam_llm_amd import model as _mm
    rows_skipped = (T - ANSWER) if (_mm.LM_HEAD_LABEL_ROWS) else 0
    d_, F_ = cfg["llm_dim"], cfg["llm_ffn"]
    skipped = 2 * 2 * rows_skipped * cfg["vocab"] * d_
    if rows_skipped and _mm.LAST_LAYER_LABEL_ROWS:
        skipped += 2 * 2 * rows_skipped * (d_ * d_ + 3 * d_ * F_)
    executed_flops = step_flops - skipped * n_clips
    ksum = timer.summary()
    gemm_all = [v for k, v in ksum.items() if k.startswith("gemm_nt")]
    gemm_ms = sum(v["total_ms"] for v in gemm_all)
    gemm_tf_all = sum(v["work"] for v in gemm_all) / (gemm_ms * 1e-3) / 1e12
    dom = max((k for k in ksum if k.startswith("gemm_nt")), key=lambda k: ksum[k]["total_ms"])
    g = ksum[dom]  # the dominant kernel (largest share of the step): one template instance of the bf16 GEMM
    gemm_tf = g["work"] / (g["total_ms"] * 1e-3) / 1e12
    kern = {k: dict(launches_per_step=v["launches"] / n_timed, ms_per_step=v["total_ms"] / n_timed,
                    avg_ms=v["avg_ms"], TFLOPs=v["work"] / (v["total_ms"] * 1e-3) / 1e12,
                    **({"algorithmic_GB_per_launch": v["bytes"] / v["launches"] / 1e9} if v.get("bytes") else {}))
            for k, v in ksum.items()}
    traffic, traffic_src = (traffic_for(dom) if args.workload == "c3" else (None, None))
    roof = {"bound": "mfma", "kernel": dom + " (slam_gemm_bf16_nt)", "achieved": gemm_tf,
            "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": gemm_tf / PEAK_BF16_TFLOPS, "traffic": None,
            "avg_launch_ms": g["avg_ms"], "launches_per_step": g["launches"] / n_timed,
            "share_of_step": g["total_ms"] / n_timed / ms_per_step,
            "timing": f"HIP events around every launch on {n_timed} of the {args.steps} timed steps (every {KERNEL_TIMING_EVERY}th, from the first): "
                      f"~1 170 events per step cost it 0.9 %, so they are not recorded on every step",
            "algorithmic_bytes_per_launch": g["bytes"] / g["launches"] if g.get("bytes") else None,
            "all_gemm_instances": {"achieved": gemm_tf_all, "share_of_step": gemm_ms / n_timed / ms_per_step}}
    pmc = pmc_for(dom) if args.workload == "c3" else None
    if pmc is not None:
        # MFMA-busy cycles / (active cycles x 256 CUs x 4 SIMDs) of the dominant kernel, longest-running shape first; at the clock the
        # chip sustained under that kernel (power-limited: 1.55-1.78 GHz of 2.4), the same kernel's TFLOP/s = mfma_util x 2.5 PF x clock / 2.4
        big = max(pmc["per_shape"].values(), key=lambda v: v["duration_us"])
        roof["mfma_util"] = big["mfma_util"]
        roof["effective_clock_GHz"] = big["effective_clock_GHz"]
        roof["pmc_detail"] = pmc
    if traffic is not None:
        hbm = traffic["fetch_bytes_per_launch"] + traffic["write_bytes_per_launch"]
        roof["traffic"] = hbm
        roof["traffic_detail"] = dict(traffic, unit="bytes per launch (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE)", source=traffic_src,
                                      ratio_to_algorithmic=(hbm / roof["algorithmic_bytes_per_launch"]) if roof["algorithmic_bytes_per_launch"] else None)
    enc_desc = {"c1": "whisper-tiny -> tinyllama-1.1b, linear projector k=5, LoRA r8",
                "c3": "whisper-large-v3 -> llama-3-8b, linear projector k=5, LoRA r16",
                "c2": "whisper-base -> llama-3-8b, linear projector k=5, LoRA r16",
                "c4": "hubert-large -> vicuna-7b, Q-Former (32 queries, 8 layers), LoRA r32"}[args.workload]
    out = {
        "metric": "audio-seconds/sec/node (Whisper-large-v3->Llama-3-8B LoRA)" if args.workload == "c3" and not args.train_encoder else
                  f"audio-seconds/sec/node ({wl['title']}: not the headline workload)",
        "value": value, "unit": "audio-seconds/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "runtime_env": {k: os.environ.get(k) for k in ("HIP_FORCE_DEV_KERNARG",)},
        "dtype": "bf16", "data": "synthetic (seeded N(0,0.1^2) audio, random token ids, random-init weights at true dims); lora_dropout 0.05 is live with "
                                  "counter-based masks, one per fused projection group: statistically, not bitwise, the reference's torch-RNG dropout "
                                  "(parity tests run at dropout 0)",
        "config": {"workload": f"{wl['title']}{' with the encoder UNFROZEN (freeze_encoder=false)' if args.train_encoder else ''}: {enc_desc} (q_proj,v_proj, dropout 0.05), "
                               f"batch {n_clips} x {clip_s:g} s clips per GPU (T={T}, {n_clips * T} frames"
                               + (" <= 12000 dynamic-frame budget" if args.workload == "c3" else "") + "), "
                               + ("GPU log-mel in the step, " if args.workload != "c4" else "raw waveform (layer-normed) in, ")
                               + "fwd+bwd+grad all-reduce+fused AdamW",
                   "global_batch_clips": world * n_clips, "seq_len": T, "parallelism": f"dp{world}",
                   "grad_exchange": None if world == 1 else ("DistributedDataParallel" if args.ddp else "GradSync (flat-buffer prefixes)"),
                   "backend": backend if backend != "nccl" else f"nccl (RCCL {rccl_version()})",
                   "n_ranks_seen": n_ranks_seen, "rccl_version": rccl_version(), "devices_distinct": devices_distinct,
                   **({"invalid_for_scaling": "ranks share devices over gloo (--functional-gloo): functional check of the N > 1 path, not a measurement"}
                      if (world > 1 and (backend != "nccl" or not devices_distinct)) else {}),
                   # max over ranks of the mean HIP-event time of GradSync.finish() per step: tail bucket launch + waits on the compute stream
                   "comm_exposed_ms": comm_exposed_ms,
                   "grad_buffer_MB": model.store.grad.numel() * 4 / 1e6 if world > 1 else None,
                   # whether the last decoder layer really ran behind its attention over the labelled rows only in this run (it does not
                   # when LoRA dropout acts on o / gate / up / down of that layer)
                   "last_layer_label_rows_active": bool(getattr(model.llm, "_last_pruned_rows", 0)),
                   # how the steps of the timed region were issued: replays of ONE captured HIP graph (forward + backward + AdamW; bit-identical
                   # to the eager step, tests/test_graph_gpu.py) or kernel by kernel from the host (the steps that carry per-launch events)
                   "step_issue": (f"{graphed.replays} graph replays + {graphed.eager_steps} eager steps over warm-up and timed region "
                                  f"(static label-row bound {model.llm.label_rows_cap})" if graphed is not None else "eager (one launch per kernel)"),
                   "logits": ("lm_head / cross entropy over the labelled rows only (the rows with label -100 enter neither loss, accuracy nor any "
                              "gradient), last decoder layer behind its attention likewise; SLAM_LM_HEAD_LABEL_ROWS=0 computes every row"
                              if rows_skipped else "full [B*T, V] lm_head computed (chunked), not materialised in fp32")},
        "loss": float(loss), "acc": float(acc),
        "model_flops_per_step_per_gpu": step_flops,
        "executed_flops_per_step_per_gpu": executed_flops,
        # mfu: the FLOPs the step executed; mfu_model: SURVEY 8(d)'s model (every row through the head) over the same time
        "mfu": executed_flops / (ms_per_step * 1e-3) / (PEAK_BF16_TFLOPS * 1e12),
        "mfu_model": step_flops / (ms_per_step * 1e-3) / (PEAK_BF16_TFLOPS * 1e12),
        "roofline": roof,
        "kernels": kern,
    }
    if world == 1 and not args.no_cpu_baseline:
        del model, opt, batch, step_model
        torch.cuda.empty_cache()
        try:
            out["cpu_baseline"] = cpu_baseline(cfg, clip_s)
        except Exception as ex:  # noqa: BLE001  (host too small etc.: report, never fake)
            out["cpu_baseline"] = {"value": None, "unit": "audio-seconds/sec", "cores": physical_cores(), "kind": "port",
                                   "sample": f"failed: {ex!r}"}
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()


if __name__ == "__main__":
    main()
