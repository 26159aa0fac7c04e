#!/usr/bin/env python
"""Fine-tune throughput of the MKGformer analogy hot path on MI355X (BASELINE.json metric).

One "step" = one full fine-tune step on one synthetic MARS-shaped batch already resident in HBM:
forward (dropout on) -> label-smoothed CE over the 2063 analogy entities + 0.43 * relaxation loss -> backward ->
(gradient all-reduce when N > 1) -> fused AdamW -> scheduler step.  Workload = BASELINE.json configs[1]:
BERT-base + ViT-B/16 patches (196 per image, 393 vision tokens), bf16 compute, batch 256 per GPU, seq_len 64.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
        bench.py --gpus 8 --steps 10 --warmup 3
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic FLOPs per example, forward; train = 3x
H, I, NH, L_TXT = 768, 3072, 12, 64


def fwd_gflop_per_example(P: int, L: int = 64, A: int = 2063) -> float:
    Nv = 1 + 2 * P
    lin = 8 * H * H + 4 * H * I
    patch = 2 * (2 * P) * H * (3 * (224 * 224 // P))
    vis_lin = Nv * lin * 12
    vis_att = sum(4 * Nv * (Nv + (L if l >= 8 else 0)) * H for l in range(12))
    txt_lin = L * lin * 12 + 4 * L * 2 * H * I
    txt_att = 12 * 4 * L * L * H
    fus = 4 * (4 * L * Nv * H)
    head = 2 * H * H + 2 * A * H
    return (patch + vis_lin + vis_att + txt_lin + txt_att + fus + head) / 1e9


def flava_fwd_gflop_per_example(P: int, L: int = 64, A: int = 2063) -> float:
    Nv, lin = 1 + 2 * P, 8 * H * H + 4 * H * I
    Sm = 1 + Nv + L
    f = 2 * (2 * P) * H * (3 * (224 * 224 // P)) + Nv * lin * 12 + 12 * 4 * Nv * Nv * H + L * lin * 12 + 12 * 4 * L * L * H
    f += 2 * (Nv + L) * H * H + Sm * lin * 6 + 6 * 4 * Sm * Sm * H + 2 * L * H * H + 2 * A * H
    return f / 1e9


def build(patch: int, seed: int, device, backbone: str = "mkgformer"):
    from mkg_analogy_amd import data_synth as D
    from mkg_analogy_amd.lit_models import TransformerLitModel
    from mkg_analogy_amd.models import FlavaKGC, MKGformerKGC, TextConfig, VisionConfig, flava_config
    torch.manual_seed(seed)
    tcfg = TextConfig()
    model = MKGformerKGC(VisionConfig(patch_size=patch), tcfg) if backbone == "mkgformer" else FlavaKGC(flava_config(patch_size=patch))
    # random-init weights of the named architecture (no checkpoints offline): N(0, 0.02) matrices / embeddings
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() >= 2 or "class_embedding" in n:
                p.normal_(0.0, 0.02)
            if "adaptive_weight.0" in n:
                p.fill_(0.25)
    cfg = D.data_config()
    args = argparse.Namespace(label_smoothing=0.1, alpha=0.43 if backbone == "mkgformer" else 0.45, pretrain=0, lr=5e-5, weight_decay=0.01,
                              optimizer="AdamW", warm_up_radio=0.1)
    lit = TransformerLitModel(model=model, args=args, tokenizer=D.FakeTokenizer(), data_config=cfg)
    model.to(device)
    lit._init_relation_word()
    return model, lit, cfg


def cpu_baseline(patch: int, L: int, iters: int = 1):
    """The CPU oracle (port of the reference algorithm, fp32, torch CPU ops) on this box's host cores, bounded sample."""
    from mkg_analogy_amd import data_synth as D
    from oracle import mkgformer_oracle as O
    # intra-op threads: torch's CPU GEMMs stop scaling (and collapse under oversubscription) well before a
    # 2-socket box's full core count at this batch size, so the baseline uses at most 32 threads and says so
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    vc = O.VisionCfg(patch_size=patch)
    tc = O.TextCfg(vocab_size=D.VOCAB)
    sd = {k: v.requires_grad_(True) for k, v in O.init_params(vc, tc, seed=0).items()}
    B = 4
    batch = D.make_batch(B, L, seed=3)
    ids = torch.tensor(D.data_config()["analogy_entity_ids"])
    live = [v for k, v in sd.items() if not k.startswith(("unimo.text_pooler", "unimo.vision_post_layernorm"))]
    opt = torch.optim.AdamW([{"params": [v for k, v in sd.items() if O.decay_of(k) > 0], "weight_decay": 0.01},
                             {"params": [v for k, v in sd.items() if O.decay_of(k) == 0], "weight_decay": 0.0}], lr=5e-5, eps=1e-8)

    def step():
        opt.zero_grad()
        _, trans = O.forward(sd, vc, tc, batch["input_ids"], batch["attention_mask"], batch["token_type_ids"], batch["pixel_values"],
                             batch["sep_idx"], train=True)
        loss, _ = O.finetune_loss(sd, trans, batch["input_ids"], batch["label"], batch["rel_idx"], batch["q_head_idx"], batch["a_head_idx"], ids, alpha=0.43)
        loss.backward()
        opt.step()
    step()
    t0 = time.time()
    for _ in range(iters):
        step()
    dt = (time.time() - t0) / iters
    return {"value": round(B / dt, 3), "unit": "examples/s", "cores": cores, "kind": "port",
            "sample": f"CPU oracle (fp32 torch) full fine-tune step, B={B}, seq_len={L}, {196 if patch == 16 else 49} patches, 1 warm-up + {iters} timed"}


class GemmTimer:
    """HIP-event timing of every gemm_nt launch (the dominant kernel) on the stream it is launched on."""

    def __init__(self, ops):
        self.ops, self.orig, self.rec = ops, ops.gemm_nt, []

    def __enter__(self):
        def timed(A, B, out, **kw):
            K = A.shape[-1] + (kw["A2"].shape[-1] if kw.get("A2") is not None else 0)
            M = kw.get("M") or (kw["a_rows"].numel() if kw.get("a_rows") is not None else A.shape[-2])
            N = kw.get("N") or (kw["b_rows"].numel() if kw.get("b_rows") is not None else B.shape[-2])
            bz = kw.get("batch", 1)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = self.orig(A, B, out, **kw)
            e.record()
            mn = float(M) * N * bz
            byt = 2.0 * M * K * bz + 2.0 * N * K * (bz if kw.get("stride_b", 0) else 1) + mn * (4 if out.dtype == torch.float32 else 2)
            for key, sz in (("preact", 2), ("mulz", 2), ("res_f32", 4), ("res_bf16", 2), ("C2", 2)):
                if kw.get(key) is not None:
                    byt += mn * sz
            t256 = ((M + 255) // 256) * ((N + 255) // 256) * bz          # same rule as mart_gemm_nt's dispatcher: >= 128 tiles and M > 128 -> 256x256 kernel
            self.rec.append((2.0 * M * N * K * bz, s, e, t256 >= 128 and M > 128 and kw.get("tile_cfg", 0) in (0, 256), byt))
            return r
        self.ops.gemm_nt = timed
        return self

    def __exit__(self, *a):
        self.ops.gemm_nt = self.orig

    def summary(self):
        torch.cuda.synchronize()
        big = [(f, s.elapsed_time(e), b) for f, s, e, isbig, b in self.rec if isbig]
        fl, ms, by = sum(x[0] for x in big), sum(x[1] for x in big), sum(x[2] for x in big)
        return len(big), fl, ms, by


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="examples per GPU")
    ap.add_argument("--seq-len", type=int, default=64)
    ap.add_argument("--patch", type=int, default=16, help="16 -> 196 patches/image (BASELINE), 32 -> 49 (reference default CLIP-B/32)")
    ap.add_argument("--model", default="mkgformer", choices=["mkgformer", "flava"], help="flava = BASELINE configs[3] (parity case; not the headline)")
    ap.add_argument("--task", default="finetune", choices=["finetune", "pretrain"],
                    help="pretrain = BASELINE configs[4]: MarKG link-prediction step, LSCE over the full 11 292-entity / 192-relation slices (use --seq-len 96)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    a = ap.parse_args()

    from mkg_analogy_amd import data_synth as D
    from mkg_analogy_amd import ops
    from mkg_analogy_amd.distributed import init_from_env
    from mkg_analogy_amd.trainer import Trainer
    import torch.distributed as dist

    rank, local, world = init_from_env()
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}"
    ops.require_gpu()
    dev = torch.device("cuda", local)
    model, lit, cfg = build(a.patch, seed=0, device=dev, backbone=a.model)
    pre = a.task == "pretrain"
    if pre:
        lit.args.pretrain = 1
    batch = D.make_batch(a.batch, a.seq_len, seed=1234 + rank, device=dev, pretrain=pre)
    total = a.steps + a.warmup + 4
    tr = Trainer(max_epochs=1, max_steps=10 * total, world_size=world)
    tr._setup(lit, [None] * (10 * total * world))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    loss = None
    for i in range(a.warmup):
        loss = tr.train_step(lit, batch, i)
    barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss = tr.train_step(lit, batch, a.warmup + i)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    ms = 1000.0 * dt / a.steps
    P = (224 // a.patch) ** 2
    value = a.batch * world * a.steps / dt
    train_gflop = 3.0 * (fwd_gflop_per_example(P, a.seq_len) if a.model == "mkgformer" else flava_fwd_gflop_per_example(P, a.seq_len))

    roof = None
    if not a.no_kernel_timing:
        # The timed region above runs the weight-gradient GEMMs (gemm_tn) and the text layers on side streams, so a gemm_nt
        # launch shares the CUs with other kernels part of the time and its start-to-end time stops being the kernel's own
        # rate.  The dominant kernel is therefore timed in one extra step with both overlaps switched off
        # (MART_OVERLAP_WGRAD=0 MART_TWO_STREAM=0 gives the same schedule for rocprofv3: profiles/*_serial*); the
        # overlapped figure is reported next to it.
        eng = model.engine
        ov, ts = getattr(eng, "overlap_wgrad", False), getattr(eng, "two_stream", False)
        with GemmTimer(ops) as gt_ov:
            tr.train_step(lit, batch, a.warmup + a.steps)
        n_ov, fl_ov, kms_ov, _ = gt_ov.summary()
        eng.overlap_wgrad = False
        eng.two_stream = False
        with GemmTimer(ops) as gt:
            tr.train_step(lit, batch, a.warmup + a.steps + 1)
        eng.overlap_wgrad, eng.two_stream = ov, ts
        n, fl, kms, by = gt.summary()
        ach = fl / (kms * 1e-3) / 1e12 if kms > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_gemm_nt.json")
        if os.path.exists(pmc) and a.batch == 256 and a.patch == 16 and a.seq_len == 64 and a.model == "mkgformer":
            traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")     # rocprofv3 --pmc passes of this same command
        roof = {"bound": "mfma", "kernel": "gemm_nt_kernel<256,256,2,4> (bf16 MFMA 32x32x16 NT GEMM, fused epilogues)", "achieved": round(ach, 1),
                "peak": 2500.0, "unit": "TFLOP/s", "frac": round(ach / 2500.0, 4), "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)",
                "algorithmic_bytes_per_launch": round(by / max(n, 1)), "launches_per_step": n,
                "avg_launch_ms": round(kms / max(n, 1), 4), "algorithmic_gflop_per_launch": round(fl / max(n, 1) / 1e9, 1),
                "step_frac_of_mfma_peak": round(value / world * train_gflop / 2.5e6, 4),
                "timing": "HIP events around every launch, one step with the side streams (weight gradients, text layers) off: kernel alone on the GPU",
                "achieved_with_wgrad_overlap": round(fl_ov / (kms_ov * 1e-3) / 1e12, 1) if kms_ov > 0 else None,
                "avg_launch_ms_with_wgrad_overlap": round(kms_ov / max(n_ov, 1), 4)}
    # Hits@1 of the (untrained, random-init) model on the same batch -- reported to exercise the ranking eval path
    metrics = tr.validate(lit, [batch])
    spread = None
    if world > 1:
        # data-parallel self-check: every replica must hold the same weights after the same all-reduced updates
        cs = model.store.master.double().sum().reshape(1)
        hi, lo = cs.clone(), cs.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        spread = float(hi - lo)

    if rank == 0:
        out = {"metric": "analogy examples/sec (fine-tune step)" if not pre else "link-prediction examples/sec (pre-train step)", "value": round(value, 2), "unit": "examples/s", "n_gpus": world,
               "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
               "config": {"workload": (f"MKGformer (BERT-base + ViT-B/{a.patch} patches)" if a.model == "mkgformer" else "FLAVA-base (12+12+6 layers)") +
                          (" fine-tune step, MARS-shaped batch" if not pre else " MarKG pre-train step (full entity / relation heads)"), "batch_per_gpu": a.batch,
                          "global_batch": a.batch * world, "seq_len": a.seq_len, "patches_per_image": P, "vision_tokens": 1 + 2 * P,
                          "entity_head": 2063 if not pre else 11292, "vocab": D.VOCAB, "parallelism": f"dp{world}", "weights": "random-init N(0,0.02)"},
               "loss": round(float(loss), 4), "hits1": metrics.get("Eval_entity/hits1"),
               "train_gflop_per_example": round(train_gflop, 1)}
        if spread is not None:
            out["replica_param_checksum_spread"] = spread        # 0.0: all ranks hold identical weights
        if roof is not None:
            out["roofline"] = roof
        if not a.no_cpu_baseline and world == 1 and a.model == "mkgformer" and not pre:
            try:
                out["cpu_baseline"] = cpu_baseline(a.patch, a.seq_len)
            except Exception as e:                      # a host-side failure of the baseline leg must not lose the GPU measurement
                out["cpu_baseline"] = {"value": None, "unit": "examples/s", "cores": min(os.cpu_count() or 1, 32), "kind": "port",
                                       "sample": f"failed: {type(e).__name__}: {e}"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
