#!/usr/bin/env python
"""bench.py — SeTok encode_images throughput on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype bf16|f32] [--workload cfg2|cfg3|cfg4|cfg4-forward|cfg5]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Started WITHOUT a launcher (`WORLD_SIZE` unset) and with --gpus N > 1, the script re-executes itself under
`torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` (the reference's own parallelism is data
parallelism under a launcher: scripts/train_setok.sh:38-39, scripts/zero2.json:16-22); it never prints `n_gpus: 1` for a run
that asked for more, and exits with an error if fewer than N GPUs are visible.

A "step" is one pass of the hot path — encode_images = ViT-L/14 tower -> +2-D pos -> DPC-kNN dynamic
clustering -> per-cluster encoder + mean -> inter-cluster encoder -> out Linear -> mm_in_projector
(mlp2x_gelu) — over ONE batch of 256 synthetic 224^2 images per GPU, bf16, inputs resident in HBM.
Images shard embarrassingly: every rank encodes its own batch, there is no collective on the data
path (weak scaling); the only collectives are the barrier and the MAX over ranks of the wall time.

Rank 0 prints ONE JSON line with the driver's fields plus `roofline` (dominant kernel = the bf16
MFMA GEMM; achieved = algorithmic FLOPs of the GEMM launches / their HIP-event durations measured
inside the timed region on the launch stream) and `cpu_baseline` (the CPU oracle — a port of the
reference's forward, validated against the reference in the build container — timed on this box's
host cores on a bounded sample of the same workload).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0          # dense bf16 MFMA peak, MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3            # fp32-input MFMA (= the fp32 vector rate), MI355X_MICROARCH.md: the parity-exact mode's GEMMs
PEAK_HBM_GBS = 8000.0              # HBM3E, MI355X_MICROARCH.md
MFMA_ONLY_RANDOM_TFLOPS = 2080.0   # measured: tools/micro/mfma_peak.hip (register-resident loop of v_mfma_f32_16x16x32_bf16 — the shape the GEMM
                                   # kernels use — on random bf16 operands, power-managed to 2.11 GHz / 1.37 kW; 2455 TFLOP/s at 2.40 GHz on
                                   # all-zero operands; the 32x32x16 shape: 1855) — profiles/r01_mfma_peak.log
B_PER_GPU, IMG, PATCH = 256, 224, 14
THRESHOLD, KNN = 0.125, 64         # dyn-k fires with seeded random-init features


T_START = time.perf_counter()


def log(msg):
    print(f"[bench +{time.perf_counter() - T_START:7.1f}s] {msg}", file=sys.stderr, flush=True)


WORKLOADS = {
    # name: (image size, images per GPU, decoder?, description)
    "cfg2": (224, 256, False, "cfg2: ViT-L/14 224^2 (24 of 24 layers, select_layer=-1 as the reference's launch scripts pass it), dyn-k DPC-kNN (k=64, threshold=0.125), SeTok head "
                               "1024/2 heads/ff 4096 -> 4096, mm_in_projector mlp2x_gelu; encode-only"),
    "cfg4-forward": (336, 128, False, "cfg4 shapes, FORWARD ONLY (the training step is not built): ViT-L/14 336^2 = 576 patches, dyn-k, batch 128 "
                                       "per GPU, same head and projector as cfg2"),
    "cfg4": (336, 128, False, "cfg4: ViT-L/14 336^2 = 576 patches (tower frozen), dyn-k, batch 128 per GPU; TRAINING STEP of the head (37.8 M "
                               "parameters: inner_encoder, inter_encoder, out): tower + head forward with saved activations, hand-written backward "
                               "from a synthetic dL/dtokens, per-module RCCL gradient all-reduce overlapped with the backward pass, AdamW on fp32 "
                               "master weights; training-mode arithmetic: the reference's proj_drop = 0.2 masks at the three dropout sites of both Blocks (seeded, regenerated in the backward pass)"),
    "cfg5": (224, 32, False, "cfg5: full Setokim forward at Vicuna-7B dims (32 layers, hidden 4096, 32 heads x 128, SwiGLU 11008, vocab 32000; random-init "
                              "bf16 weights): 32 images -> SeTok encode (cfg2 model) -> mm_in_projector -> splice into 512-token prompts -> LLM prefill -> "
                              "logits at every position -> language-model loss over the answer part (setokim_llama.py:94-160; the diffusion term is out of scope)"),
    "cfg3": (224, 256, True, "cfg3: cfg2 encode + reconstruction decoder (SetokDeTokenizer: token_feat_dim 4096 -> Q-Former 768/12 heads/6 layers, "
                              "324 queries (image_size 256 / 14), cross-attention every 2nd layer -> 16 x ViT block 768/16 heads -> LayerNorm -> "
                              "to_pixels (768 -> 14*14*3) -> unpatchify to 252x252 -> mean-squared reconstruction error against a synthetic gold image "
                              "(the pixel head the reference leaves undefined, detokenizer.py:101-120; its GAN / LPIPS terms are out of scope)"),
}


def build_decoder(device):
    import setok_amd
    det = setok_amd.SetokDeTokenizer(token_feat_dim=4096, hidden_dim=768, patch_size=14, image_size=256, decoder_embed_dim=768,
                                     decoder_nheads=16, decoder_depth=16, feature_mapper_path_or_name="bert-base-uncased", pixel_head=True)
    return det.to(device=device, dtype=torch.bfloat16).eval()


def build_model(device, img=IMG, dtype=torch.bfloat16, select_layer=-1):
    import setok_amd
    from setok_amd.synthetic import init_synthetic_
    vit = dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16,
               image_size=img, patch_size=PATCH)
    tok = setok_amd.SetokTokenizer(vision_tower=vit, mm_vision_select_layer=select_layer, hidden_dim=1024, token_feat_dim=4096,
                                   min_cluster_num=64, threshold=THRESHOLD, nheads=2, dim_feedforward=4096)
    init_synthetic_(tok, tower_seed=0, head_seed=1)
    proj = setok_amd.build_vision_projector("mlp2x_gelu", mm_hidden_size=4096, hidden_size=4096)
    torch.manual_seed(2)
    for m in proj:
        if isinstance(m, torch.nn.Linear):
            torch.nn.init.xavier_uniform_(m.weight); torch.nn.init.zeros_(m.bias)
    return tok.to(device=device, dtype=dtype).eval(), proj.to(device=device, dtype=dtype).eval()


def cpu_baseline(tok, proj, n_images=16, reps=3, select_layer=-1):
    """Oracle (oracle/setok_oracle.py) on the host cores: fp32, same weights (upcast from the bf16 model),
    same synthetic image distribution, micro-batch of `n_images`."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import setok_oracle as O
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 32))               # more threads than this only adds sync overhead to the small per-cluster ops
    torch.set_num_threads(cores)
    sd = {k: v.detach().float().cpu() for k, v in tok.state_dict().items()}
    psd = {k: v.detach().float().cpu() for k, v in proj.state_dict().items()}
    vc, hc = O.VitConfig(), O.HeadConfig(threshold=THRESHOLD, mm_vision_select_layer=select_layer)     # the same layer selection as the timed GPU steps
    g = torch.Generator().manual_seed(3)
    images = torch.randn(n_images, 3, IMG, IMG, generator=g)
    t0 = time.perf_counter()
    O.encode_images(sd, psd, "mlp2x_gelu", vc, hc, images[:2])          # warm-up, also sizes the sample
    warm = time.perf_counter() - t0
    if warm * n_images * reps / 2 > 60.0:                               # keep the leg bounded (~10-30 s of CPU work)
        reps = 1
        n_images = max(2, min(n_images, int(30.0 / (warm / 2))))
        images = images[:n_images]
    log(f"cpu_baseline: warm-up 2 images {warm:.1f} s on {cores} threads; timing {reps} x {n_images}")
    t0 = time.perf_counter()
    for _ in range(reps):
        out = O.encode_images(sd, psd, "mlp2x_gelu", vc, hc, images)
    dt = time.perf_counter() - t0
    return dict(value=round(n_images * reps / dt, 3), unit="images/s", cores=cores, kind="port",
                sample=f"{reps} x {n_images} images of the same workload (select_layer {select_layer}, fp32, torch CPU, {cores} threads, {dt:.1f} s); "
                       f"tokens/img {sum(o.shape[0] for o in out) / n_images:.1f}")


def gpu_telemetry(step, device_index, n_steps=6):
    """Shader clock and socket power while the workload runs (rocm-smi sampled from a side thread during a few extra, untimed
    steps).  The chip is power-capped on this workload, so the clock it sustains — not the 2.4 GHz of the datasheet peak — sets
    the reachable MFMA rate; both fractions are reported.  Returns {} if rocm-smi is unavailable."""
    import re, subprocess, threading
    samples, stop = [], [False]

    def sampler():
        while not stop[0]:
            try:
                out = subprocess.run(["rocm-smi", "-d", str(device_index), "--showclocks", "--showpower"], capture_output=True,
                                     text=True, timeout=10).stdout
                m = re.search(r"sclk clock level:\s*\d+:\s*\((\d+)Mhz\)", out)
                w = re.search(r"Power \(W\):\s*([0-9.]+)", out)
                if m:
                    samples.append((int(m.group(1)), float(w.group(1)) if w else None))
            except Exception:
                return
            time.sleep(0.05)
    th = threading.Thread(target=sampler, daemon=True)
    torch.cuda.synchronize()
    th.start()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.5 or len(samples) < 3:
        for _ in range(n_steps):
            step()
        torch.cuda.synchronize()
        if time.perf_counter() - t0 > 8.0:
            break
    stop[0] = True
    th.join(timeout=15)
    samples = samples[1:] if len(samples) > 1 else samples          # the first sample may predate the load
    if not samples:
        return {}
    clk = sorted(s[0] for s in samples)[len(samples) // 2]
    pw = [s[1] for s in samples if s[1] is not None]
    return {"sclk_mhz_under_load": clk, "socket_power_w_under_load": round(sorted(pw)[len(pw) // 2], 0) if pw else None}


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: become N ranks.  Re-executes this script under torch.distributed.run (one process per
    GPU, rendezvous on 127.0.0.1 at a free port); the re-executed ranks see WORLD_SIZE and take the normal path."""
    import socket
    if not args.launch_check and not args.share_gpu:
        n_vis = torch.cuda.device_count()
        if n_vis < args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} asked for, {n_vis} GPU(s) visible: refusing to report fewer ranks than asked")
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("no launcher in the environment: " + " ".join(cmd))
    sys.stderr.flush()
    os.execv(sys.executable, cmd)


def launch_check(rank, world, local):
    """The launcher and the multi-rank bookkeeping on CPU (gloo): every rank reports itself, rank 0 prints what a bench line would carry.
    Used by tests/test_parallel_cpu.py (`python bench.py --gpus 2 --launch-check`)."""
    import torch.distributed as dist
    from setok_amd.parallel import max_over_ranks
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    dt = max_over_ranks(0.01 * (rank + 1))
    per_rank = [None] * world
    if world > 1:
        dist.all_gather_object(per_rank, dict(rank=rank, local_rank=local, seconds=0.01 * (rank + 1)))
    else:
        per_rank = [dict(rank=0, local_rank=0, seconds=0.01)]
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "ranks": per_rank, "slowest_seconds": dt}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def load_traffic(applies):
    """HBM bytes per launch from the PMC passes of this command (profiles/rNN_traffic.json, written by tools/gemm_traffic.py from
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs; the newest round's file).  A missing file is reported as such; a file that is there
    and cannot be read is an error — evidence is never dropped silently."""
    import glob
    if not applies:                              # the PMC passes are of the default command (cfg2, bf16, batch 256): other workloads launch other shapes
        return None, "the PMC passes (profiles/rNN_traffic.json) are of the default workload (cfg2, bf16, 256 images); not attributed to this one"
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_traffic.json")))
    if not files:
        return None, "no profiles/rNN_traffic.json in the tree"
    with open(files[-1]) as f:
        d = json.load(f)
    for key in ("gemm", "clustering"):
        if key not in d or "traffic_bytes_per_launch" not in d[key]:
            raise KeyError(f"{files[-1]}: no {key}.traffic_bytes_per_launch")
    d["file"] = os.path.relpath(files[-1], ROOT)
    return d, None


def live_traffic(select_layer, log):
    """The two PMC passes of THIS command on THIS box (round 5; rounds 2-4 replayed the builder's committed passes into the driver's line):
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` as separate sub-processes of `bench.py --steps 2 --warmup 1 --timed-only`
    (counters cannot be collected inside the timed run; kernel-trace only beside them), reduced by tools/gemm_traffic.py exactly as
    tools/profile_round.sh does.  Returns (dict, None) or (None, why) — the caller then falls back to the committed file and says so."""
    import shutil, subprocess, tempfile
    if not shutil.which("rocprofv3"):
        return None, "rocprofv3 is not on PATH"
    tmp = tempfile.mkdtemp(prefix="setok_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    env.pop("RANK", None); env.pop("WORLD_SIZE", None); env.pop("LOCAL_RANK", None)
    dbs = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            t0 = time.perf_counter()
            r = subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", counter, "-d", out, "-o", "p", "--", sys.executable, os.path.join(ROOT, "bench.py"),
                                "--steps", "2", "--warmup", "1", "--timed-only", "--select-layer", str(select_layer)],
                               cwd="/tmp", env=env, capture_output=True, text=True, timeout=150)
            found = [os.path.join(dp, f) for dp, _, fs in os.walk(out) for f in fs if f.endswith(".db")]
            if r.returncode != 0 or not found:
                return None, f"the {counter} pass failed (rc {r.returncode}): {(r.stderr or r.stdout)[-200:]}"
            dbs[counter] = found[0]
            log(f"PMC pass {counter}: {time.perf_counter() - t0:.1f} s")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gemm_traffic.py"), dbs["FETCH_SIZE"], dbs["WRITE_SIZE"]],
                           capture_output=True, text=True, timeout=120)
        if r.returncode != 0:
            return None, f"tools/gemm_traffic.py failed: {(r.stderr or r.stdout)[-200:]}"
        d = json.loads(r.stdout)
        d["file"] = "live: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE passes of this command on this box, inside this run"
        return d, None
    except Exception as e:                                            # a profiler that is missing, refuses or hangs must not cost the bench line
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def gemm_class(p):
    return p["kernel"].split(":", 1)[1] if ":" in p["kernel"] else "all"


def main():
    # Every workload here is a forward pass (cfg4 adds the hand-written training step, which needs no autograd graph either).  Since round 3 the
    # modules follow torch's rule — gradients enabled + a parameter requiring one => a graph is recorded or the call is refused — so the benchmark
    # says what it measures: the frozen / no_grad path, as the reference's inference callers run it.
    torch.set_grad_enabled(False)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--dtype", choices=["bf16", "f16", "f32"], default="bf16",
                    help="bf16 = the BASELINE metric's precision; f16 = what the reference's inference loader casts the tower to (src/model/builder.py:135-136; "
                         "libsetok_hip_f16.so, the same MFMA rate); f32 = the parity-exact mode (bit-exact cluster indices, 1e-4 features)")
    ap.add_argument("--select-layer", type=int, default=-1,
                    help="hidden_states index the tower returns: -1 (default) = what the reference's launch scripts pass and its training dataclass "
                         "defaults to (scripts/pretrain_mm_proj.sh:43, scripts/finetune.sh:67, src/train/training_utils.py:25: all 24 layers run); "
                         "-2 = the reference classes' own default (tokenizer.py:18, 23 of 24 layers), reported beside it as `also_select_layer_minus2`")
    ap.add_argument("--probe-every", type=int, default=4,
                    help="HIP events ride on every GEMM launch of every N-th timed step (the first one included); the steps in between run unprobed. An event pair "
                         "costs ~4 us of device time per launch (0.43 ms of a 42 ms step when every step is probed: N = 1, rounds 1-4)")
    ap.add_argument("--no-live-traffic", action="store_true", help="skip the two rocprofv3 PMC sub-runs behind the timed region (≈ 1 min); the committed passes are quoted instead")
    ap.add_argument("--timed-only", action="store_true",
                    help="profiling runs: nothing but warm-up + the timed steps (no clock / power sampling loop, no select_layer = -2 steps, no CPU "
                         "baseline), so that two rocprofv3 passes of the same command see the same launches")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="nccl (= RCCL over xGMI) is the measured configuration; gloo + --share-gpu run the multi-rank logic on a one-GPU box (validation only)")
    ap.add_argument("--share-gpu", action="store_true", help="validation only: every rank uses cuda:0 (RCCL refuses two ranks on one device: use --backend gloo)")
    ap.add_argument("--launch-check", action="store_true", help="CPU-only check of the self-launcher and the rank bookkeeping (gloo)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="cfg2",
                    help="cfg2 is the BASELINE.json metric; the others are additional measurements (no cpu_baseline leg)")
    args = ap.parse_args()
    img, b_default, with_decoder, workload_desc = WORKLOADS[args.workload]
    if args.batch is None:
        args.batch = b_default
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)                                   # does not return

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}: the launcher and the flag disagree")
    if args.launch_check:
        return launch_check(rank, world, local)
    dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype]
    if args.share_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    numa = {}
    if world > 1 and not args.share_gpu:
        from setok_amd.parallel import pin_host_to_gpu_numa_node
        numa = pin_host_to_gpu_numa_node(local)                     # this rank's host threads next to its GPU (best effort; reported per rank)
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    import setok_amd
    from setok_amd import ops
    log(f"rank {rank}/{world}: building model")
    if args.select_layer != -1:
        n_run = 24 + 1 + args.select_layer if args.select_layer < 0 else args.select_layer
        workload_desc = workload_desc.replace("24 of 24 layers, select_layer=-1 as the reference's launch scripts pass it",
                                              f"{n_run} of 24 layers, select_layer={args.select_layer}")
    if args.dtype == "f32":
        workload_desc += "; fp32 parity mode (exact-f32 MFMA GEMMs)"
    tok, proj = build_model(dev, img, dtype, args.select_layer)
    det = build_decoder(dev) if with_decoder else None
    log("model on device")
    B = args.batch
    g = torch.Generator().manual_seed(3 + rank)
    images = torch.randn(B, 3, img, img, generator=g).to(device=dev, dtype=dtype)   # resident in HBM

    llm = None
    if args.workload == "cfg5":
        from setok_amd.llama import SetokimLlamaPrefill
        lcfg = dict(vocab_size=32000, hidden_size=4096, intermediate_size=11008, num_hidden_layers=32, num_attention_heads=32,
                    num_key_value_heads=32, rms_norm_eps=1e-5, rope_theta=10000.0)
        with torch.device(dev):
            llm = SetokimLlamaPrefill(lcfg, vision_tower=tok, mm_in_projector=proj).to(dtype)
        gl = torch.Generator(device=dev).manual_seed(11)
        for n_, p_ in llm.named_parameters():
            if n_.startswith(("vision_tower.", "mm_in_projector.")):
                continue
            if p_.dim() == 2:
                p_.data.normal_(0.0, 0.02, generator=gl)
            else:
                p_.data.fill_(1.0)
        llm.eval()
        T_TXT = 512
        ids = torch.randint(0, 32000, (B, T_TXT), generator=torch.Generator().manual_seed(5 + rank))
        ids[:, 17] = -200                                         # one image placeholder per prompt (IMAGE_TOKEN_INDEX)
        ids = ids.to(dev)
        amask = torch.ones(B, T_TXT, dtype=torch.bool, device=dev)
        lm_labels = ids.clone()
        lm_labels[:, :64] = -100                                  # the prompt part (IGNORE_INDEX); the image positions are set by the splice
        log("LLM on device")
    trainer = None
    if args.workload == "cfg4":
        from setok_amd.training import HeadTrainer
        trainer = HeadTrainer(tok, lr=1e-5, weight_decay=0.0, dropout="train", dropout_seed=1234)    # the reference's training objective: proj_drop masks active

    def step():
        if llm is not None:
            logits, _, _, loss = llm(input_ids=ids, attention_mask=amask, labels=lm_labels, comp_images=images, return_loss=True)
            step.logits, step.loss = logits, loss
            return llm._last_features
        if trainer is not None:
            tokens, ctx = trainer.forward(images)
            trainer.backward(ctx, tokens.packed * 1e-3)          # dL/dtokens of L = 5e-4 |tokens|^2, standing in for the projector / LLM
            trainer.step()
            return tokens
        if det is None:
            return setok_amd.encode_images(tok, proj, images)
        tokens, _, _ = tok(images)                   # SeTok.forward (src/model/setok/model.py:87-88): tokenize, then detokenize
        step.recon_loss = det.reconstruction_loss(tokens, step.gold)     # decoder -> to_pixels -> unpatchify -> MSE (model.py:75-76,91)
        return tokens

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if det is not None:                          # the gold image the reconstruction is scored against (ImageNet stand-in at the decoder's 18 x 14 = 252 px)
        side = det.height * det.patch_size
        step.gold = torch.randn(B, 3, side, side, generator=torch.Generator().manual_seed(7 + rank)).to(device=dev, dtype=dtype)

    for i in range(args.warmup):
        out = step()
        torch.cuda.synchronize()
        log(f"warmup step {i} done")
    startup_s = time.perf_counter() - T_START                       # process start -> model built, context warm, warm-up steps done: what a rank spends before the barrier
    barrier()
    ops.profile_start()                          # HIP events around every GEMM launch, on the launch stream ...
    pe = max(1, args.probe_every)                # ... of every pe-th timed step: the probe has a cost of its own (see --probe-every)
    probed_steps = 0
    t0 = time.perf_counter()
    for i in range(args.steps):
        if pe > 1:
            ops.profile_pause(i % pe != 0)
        probed_steps += 1 if i % pe == 0 else 0
        out = step()
    barrier()
    dt_local = time.perf_counter() - t0
    prof = ops.profile_stop()
    log(f"timed {args.steps} steps in {dt_local:.3f} s")
    from setok_amd.parallel import max_over_ranks
    dt = max_over_ranks(dt_local, device=dev)            # wall time of the slowest rank

    counts = out.counts
    mine = dict(rank=rank, ms_per_step=round(dt_local / args.steps * 1e3, 3), tokens_per_image=round(sum(counts) / len(counts), 2),
                startup_s=round(startup_s, 1), **({"numa_node": numa.get("numa_node"), "host_cpus": numa.get("cpus")} if numa else {}))
    if trainer is not None:
        mine.update(trainer.comm_stats())
    per_rank = [mine]
    if dist is not None:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    # the select_layer = -2 number (the reference classes' default, 23 layers) beside the headline's -1: a few extra steps, untimed above
    other = None
    if rank == 0 and args.workload == "cfg2" and world == 1 and args.select_layer == -1 and not args.timed_only:
        tower = tok.image_feature_encoder
        tower.select_layer = -2
        setok_amd.encode_images(tok, proj, images); torch.cuda.synchronize()
        n_o = max(2, min(args.steps, 5))
        t1 = time.perf_counter()
        for _ in range(n_o):
            o2 = setok_amd.encode_images(tok, proj, images)
        torch.cuda.synchronize()
        d1 = (time.perf_counter() - t1) / n_o
        other = {"select_layer": -2, "layers_run": 23, "images_per_s": round(B / d1, 2), "ms_per_step": round(d1 * 1e3, 3), "steps": n_o,
                 "tokens_per_image_mean": round(sum(o2.counts) / len(o2.counts), 2)}
        tower.select_layer = -1

    # clock / power under load: rank 0 re-runs the step while polling rocm-smi — only where the step has no collective (a training step's
    # all-reduce would wait for ranks that are not stepping)
    telemetry = gpu_telemetry(step, local) if rank == 0 and not args.timed_only and (world == 1 or trainer is None) else None
    applies = args.workload == "cfg2" and args.dtype == "bf16" and B == 256 and args.select_layer == -1
    traffic, traffic_note = None, None
    if rank == 0 and world == 1 and applies and not args.timed_only and not args.no_cpu_baseline and not args.no_live_traffic:
        traffic, why = live_traffic(args.select_layer, log)           # the default run: measured here, on this box
        if traffic is None:
            log(f"live PMC passes unavailable ({why}); falling back to the committed passes")
            live_note = why
        else:
            live_note = None
    else:
        live_note = "not collected in this invocation (multi-rank, --timed-only, --no-cpu-baseline or --no-live-traffic)"
    if traffic is None:
        traffic, traffic_note = load_traffic(applies)
        if traffic is not None:
            traffic["file"] = f"{traffic['file']} (the builder's committed passes; live passes: {live_note})" 
    if rank == 0:
        gname = {"bf16": "gemm_bf16", "f16": "gemm_f16", "f32": "gemm_f32"}[args.dtype]
        peak = PEAK_F32_TFLOPS if args.dtype == "f32" else PEAK_BF16_TFLOPS                    # (fp16 and bf16 MFMA: the same dense peak)
        gemm = [p for p in prof if p["kernel"].split(":")[0] == gname]
        g_ms = sum(p["ms"] for p in gemm)
        g_fl = sum(p["flops"] for p in gemm)
        achieved = g_fl / (g_ms * 1e-3) / 1e12 if g_ms > 0 else 0.0
        classes = {}
        for p_ in gemm:
            c = classes.setdefault(gemm_class(p_), dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            c["launches"] += 1; c["ms"] += p_["ms"]; c["flops"] += p_["flops"]; c["bytes"] += p_.get("bytes", 0.0)
        per_class = {k: {"launches_per_step": v["launches"] // max(probed_steps, 1), "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1),
                         "avg_launch_ms": round(v["ms"] / v["launches"], 4),
                         "algorithmic_mb_per_launch": round(v["bytes"] / v["launches"] / 1e6, 1)} for k, v in sorted(classes.items())}
        alg_bytes = sum(p_.get("bytes", 0.0) for p_ in gemm) / max(len(gemm), 1)
        res = {
            "metric": "images/s SeTok encode (ViT-L/14, 224^2, dyn-k)" if args.workload == "cfg2" else f"images/s SeTok {args.workload}",
            "value": round(world * B * args.steps / dt, 2),
            "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "probe_every": pe,                              # measurement mode of value / ms_per_step: 1 = rounds 1-4's (every step probed), 4 = default since round 5
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": workload_desc, "batch_per_gpu": B, "global_batch": world * B,
                       "tokens_per_image": {"mean": round(sum(counts) / len(counts), 2), "min": min(counts), "max": max(counts)},
                       "sharding": (f"dp{world} (images sharded; the head's gradients are all-reduced per module in flat buckets, overlapped with the backward pass)"
                                    if trainer is not None else f"dp{world} (images sharded, no data-path collective)") + ("" if args.backend == "nccl" and not args.share_gpu else
                                                                                                   f" [VALIDATION RUN: backend {args.backend}, ranks share one GPU: not a scaling number]"),
                       "per_rank": per_rank, "slowest_rank": max(per_rank, key=lambda r: r["ms_per_step"])["rank"]},
            "roofline": {"bound": "mfma", "kernel": "gemm_pp_kernel<*> / gemm_persist_kernel<*> / gemm_tail_kernel<*> (every 16-bit MFMA GEMM launch of the step)" if args.dtype != "f32"
                         else "gemm_f32_kernel (exact-f32 MFMA GEMM, parity mode)",
                         "achieved": round(achieved, 1), "peak": peak,
                         "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                         "traffic": traffic["gemm"]["traffic_bytes_per_launch"] if traffic else None,
                         "algorithmic_bytes_per_launch": round(alg_bytes),
                         "launches_per_step": len(gemm) // max(probed_steps, 1),
                         "probe": f"HIP events on every GEMM launch of {probed_steps} of the {args.steps} timed steps (every {pe}-th; --probe-every)",
                         "avg_launch_ms": round(g_ms / max(len(gemm), 1), 4),
                         "avg_launch_gflop": round(g_fl / max(len(gemm), 1) / 1e9, 2),
                         "gemm_share_of_step": round(g_ms / max(probed_steps, 1) / (dt / args.steps * 1e3), 3),
                         "per_class": per_class},
        }
        if traffic:
            res["roofline"]["traffic_source"] = traffic["file"]
            res["roofline"]["traffic_per_class"] = traffic["gemm"].get("per_class")
        else:
            res["roofline"]["traffic_note"] = traffic_note
        clus = [p for p in prof if p["kernel"] == "cluster_dpc_knn"]
        if clus:
            c_ms = sum(p["ms"] for p in clus) / len(clus)
            c_bytes = sum(p["flops"] for p in clus) / len(clus)                              # the "flops" slot holds algorithmic BYTES here
            c_gbs = c_bytes / (c_ms * 1e-3) / 1e9
            gram_tf = 2.0 * B * (img // PATCH) ** 4 * 1024 / (c_ms * 1e-3) / 1e12           # the Gram's 2 N^2 C per image over the whole call
            res["roofline_clustering"] = {"bound": "hbm", "kernel": "setok_cluster_dpc_knn (Gram -> kNN density -> delta/score -> centres -> assignment)",
                                          "achieved": round(c_gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(c_gbs / PEAK_HBM_GBS, 4),
                                          "algorithmic_bytes_per_call": round(c_bytes),
                                          "traffic": traffic["clustering"]["traffic_bytes_per_launch"] if traffic else None,
                                          "gram_tflops_over_the_call": round(gram_tf, 1),
                                          "ms_per_call": round(c_ms, 4), "share_of_step": round(c_ms * (len(clus) / max(probed_steps, 1)) / (dt / args.steps * 1e3), 4)}
        if args.dtype != "f32":
            res["roofline"]["mfma_only_random_operands_tflops"] = MFMA_ONLY_RANDOM_TFLOPS
            res["roofline"]["frac_of_mfma_only_random"] = round(achieved / MFMA_ONLY_RANDOM_TFLOPS, 4)
        if telemetry:
            res["roofline"].update(telemetry)
            if telemetry.get("sclk_mhz_under_load"):
                pk = peak * telemetry["sclk_mhz_under_load"] / 2400.0
                res["roofline"]["peak_at_measured_clock"] = round(pk, 1)
                res["roofline"]["frac_at_measured_clock"] = round(achieved / pk, 4)
        if other:
            res["config"]["also_select_layer_minus2"] = other
        if det is not None:
            res["config"]["reconstruction_mse"] = round(float(step.recon_loss), 6)      # the step's terminal scalar (random-init decoder vs a random gold image)
        if llm is not None:
            res["config"]["lm_loss"] = round(float(step.loss), 6)
        if not args.no_cpu_baseline and not args.timed_only and world == 1 and args.workload == "cfg2":
            res["cpu_baseline"] = cpu_baseline(tok, proj, select_layer=args.select_layer)
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
