#!/usr/bin/env python
"""Fine-tune throughput of the MKGformer analogy hot path on MI355X (BASELINE.json metric).

One "step" = one full fine-tune step on one synthetic MARS-shaped batch already resident in HBM:
forward (dropout on) -> label-smoothed CE over the scored entity head (--entity-head: all 11292 MarKG entities by default = north_star's
"~11k-entity head"; 2063 = the MARS analogy entities of the reference's fine-tune branch) + 0.43 * relaxation loss -> backward ->
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


def build(patch: int, seed: int, device, backbone: str = "mkgformer", entity_head: int = 2063):
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
    if entity_head == D.N_ENT:       # north_star's "~11k-entity head": the fine-tune step scores the [MASK] row against EVERY MarKG entity
        cfg["analogy_entity_ids"] = list(range(D.BASE_VOCAB, D.BASE_VOCAB + D.N_ENT))
    else:
        assert entity_head == D.N_ANALOGY, "entity head: 2063 (MARS analogy entities, lit_models/transformer.py:95) or 11292 (all MarKG entities)"
    args = argparse.Namespace(label_smoothing=0.1, alpha=0.43 if backbone == "mkgformer" else 0.45, pretrain=0, lr=5e-5, weight_decay=0.01,
                              optimizer="AdamW", warm_up_radio=0.1)
    lit = TransformerLitModel(model=model, args=args, tokenizer=D.FakeTokenizer(), data_config=cfg)
    model.to(device)
    lit._init_relation_word()
    return model, lit, cfg


def cpu_step_rate(patch: int, L: int, threads: int, iters: int = 3, B: int = 8) -> float:
    """examples/s of the CPU oracle's full fine-tune step (forward, loss, backward, AdamW) on ``threads`` host threads."""
    from mkg_analogy_amd import data_synth as D
    from oracle import mkgformer_oracle as O
    torch.set_num_threads(threads)
    vc = O.VisionCfg(patch_size=patch)
    tc = O.TextCfg(vocab_size=D.VOCAB)
    batch = D.make_batch(B, L, seed=3)
    ids = torch.tensor(D.data_config()["analogy_entity_ids"])
    sd = {k: v.requires_grad_(True) for k, v in O.init_params(vc, tc, seed=0).items()}
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
    return B * iters / (time.time() - t0)


def usable_cpus() -> int:
    """Cores this process may actually use: the affinity mask, capped by the cgroup CPU quota (the GPU boxes show 256 logical cores
    and run the container under ``cpu.max = 1600000 100000``: 16 cores' worth of time, whatever the thread count)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(patch: int, L: int, iters: int = 3, B: int = 8):
    """The CPU oracle (port of the reference algorithm, fp32, torch CPU ops) on ALL host cores this process may use, as SURVEY 8(d)
    defines it: the same synthetic batch shape at B=8, one warm-up + three timed full fine-tune steps.  ``cores`` = the threads
    used = the cgroup CPU quota of the container (16 on the GPU boxes, which list 256 logical cores: wider pools only get the same
    16 cores' worth of time and thrash -- measured 2.61 examples/s on 16 threads, 2.05 on 32, 0.96 on 64, 0.44 on 128)."""
    n = usable_cpus()
    return {"value": round(cpu_step_rate(patch, L, n, iters, B), 3), "unit": "examples/s", "cores": n, "kind": "port",
            "sample": f"CPU oracle (fp32 torch) full fine-tune step (fwd + loss + bwd + AdamW), B={B}, seq_len={L}, {196 if patch == 16 else 49} patches, "
                      f"1 warm-up + {iters} timed, torch.set_num_threads({n}) = every core the container may use "
                      f"(host lists {os.cpu_count()} logical cores, cgroup quota / affinity allow {n})"}


class KernelTimer:
    """HIP-event timing of every launch of the three MFMA kernel families (on the stream each one is launched on): gemm_nt (the
    dominant kernel), gemm_tn (weight gradients) and the attention kernels -- and of the largest HBM-bound family, LayerNorm forward /
    backward, with its algorithmic bytes (every operand once)."""

    def __init__(self, ops):
        self.ops, self.orig, self.rec = ops, {n: getattr(ops, n) for n in ("gemm_nt", "gemm_tn", "attn_fwd", "attn_bwd", "ln_fwd", "ln_bwd")}, []

    def _wrap(self, name, work):
        orig = self.orig[name]

        def timed(*a, **kw):
            fam, fl, byt = work(*a, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = orig(*a, **kw)
            e.record()
            self.rec.append((fam, fl, s, e, byt))
            return r
        return timed

    def __enter__(self):
        def nt(A, B, out, **kw):
            K = A.shape[-1] + (kw["A2"].shape[-1] if kw.get("A2") is not None else 0)
            M = kw.get("M") or (kw["a_rows"].numel() if kw.get("a_rows") is not None else A.shape[-2])
            N = kw.get("N") or (kw["b_rows"].numel() if kw.get("b_rows") is not None else B.shape[-2])
            bz = kw.get("batch", 1)
            mn = float(M) * N * bz
            byt = 2.0 * M * K * bz + 2.0 * N * K * (bz if kw.get("stride_b", 0) else 1) + mn * (4 if out.dtype == torch.float32 else 2)
            for key, sz in (("preact", 2), ("mulz", 2), ("res_f32", 4), ("res_bf16", 2), ("C2", 2)):
                if kw.get(key) is not None:
                    byt += mn * sz
            t256 = ((M + 255) // 256) * ((N + 255) // 256) * bz          # same rule as mart_gemm_nt's dispatcher: >= 128 tiles and M > 128 -> 256x256 kernel
            big = t256 >= 128 and M > 128 and kw.get("tile_cfg", 0) in (0, 256)
            return ("gemm_nt_256" if big else "gemm_nt_128"), 2.0 * M * N * K * bz, byt

        def tn(X, Y, out, **kw):
            M = kw.get("M") or X.shape[-2]
            NX = kw.get("NX") or X.shape[-1]
            NY = kw.get("NY") or Y.shape[-1]
            bz = kw.get("batch", 1)
            return "gemm_tn", 2.0 * M * NX * NY * bz, (2.0 * M * (NX + NY) + 8.0 * NX * NY) * bz

        def att(mult):
            def f(**kw):
                S = kw["Sq"] * (kw["Sk"] + kw.get("Lp", 0))
                fam = ("attn_fwd" if mult == 1 else "attn_bwd") + ("_vision" if kw["Sq"] > 128 else "_text")
                return fam, mult * 4.0 * kw["B"] * kw["nh"] * S * 64, 0.0
            return f
        def ln(which, sizes):
            def f(**kw):
                n = float(kw["M"]) * kw["H"]
                return which + ("_vision" if kw["M"] > 50000 else "_text"), 0.0, sum(n * sz for key, sz in sizes if kw.get(key) is not None)
            return f
        self.ops.ln_fwd = self._wrap("ln_fwd", lambda **kw: ln("ln_fwd", (("x_f32", 4), ("y_bf16", 2), ("s_out", 4), ("out_f32", 4), ("out_bf16", 2)))(**kw))
        self.ops.ln_bwd = self._wrap("ln_bwd", lambda **kw: ln("ln_bwd", (("dy_f32", 4), ("dy_bf16", 2), ("s", 4), ("add_f32", 4), ("add_bf16", 2), ("ds_f32", 4), ("ds_bf16", 2)))(**kw))
        self.ops.gemm_nt = self._wrap("gemm_nt", nt)
        self.ops.gemm_tn = self._wrap("gemm_tn", tn)
        self.ops.attn_fwd = self._wrap("attn_fwd", lambda **kw: att(1)(**kw))
        self.ops.attn_bwd = self._wrap("attn_bwd", lambda **kw: att(2)(**kw))
        return self

    def __exit__(self, *a):
        for n, f in self.orig.items():
            setattr(self.ops, n, f)

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for fam, fl, s, e, byt in self.rec:
            d = out.setdefault(fam, [0, 0.0, 0.0, 0.0])
            d[0] += 1; d[1] += fl; d[2] += s.elapsed_time(e); d[3] += byt
        return out


def source_sha16(*names):
    import hashlib
    h = hashlib.sha256()
    for n in names:
        h.update(open(os.path.join(ROOT, "mkg_analogy_amd", "csrc", n), "rb").read())
    return h.hexdigest()[:16]


def reference_parity(model, lit, batch, cfg, dev, timed_weights: str):
    """The timed code path (same engine, same precision configuration) AND the evaluation default (fp32-accurate pass, TransformerLitModel._eval_at)
    against the UNMODIFIED reference: eval-mode logits of the [MASK] rows of the first 32 examples of the rank-0 bench batch over the 2063 MARS
    analogy entities, compared with tests/golden/g7_bench_{cond,plain}.npz (written by oracle/gen_goldens_full.py from /root/reference; same batch
    seed, weights regenerated from the goldens' numpy seed).  One verdict PER WEIGHT SET (never one flag for the line): ``cond`` is the set the
    default run times.  Runs LAST: it overwrites the timed network's weights."""
    import numpy as np
    from mkg_analogy_amd import data_synth as D
    keys = ("input_ids", "attention_mask", "token_type_ids", "pixel_values", "sep_idx")
    ids = torch.tensor(D.data_config(seed=1234)["analogy_entity_ids"], device=dev)
    res = {"what": "eval-mode [MASK]-row logits vs the unmodified reference CPU path (goldens G7: first 32 examples of this batch x 2063 analogy entities, weights "
                   "regenerated from the goldens' seed), for the TIMED bf16 configuration and for the fp32-accurate evaluation default; north_star: 1e-2 (bf16) / "
                   "1e-3 (fp32) on logits, bit-exact ranked entity indices",
           "timed_mode": f"vision bf16, text_f16={int(model.engine.text_f16)}, head_split={int(model.engine.head_split)}, last_layer_rows={int(lit.last_layer_rows)}",
           "timed_weights": timed_weights, "north_star_tol": {"bf16": 1e-2, "fp32": 1e-3}}
    model.eval()
    for tag, cond in (("g7_bench_cond", True), ("g7_bench_plain", False)):
        path = os.path.join(ROOT, "tests", "golden", tag + ".npz")
        if not os.path.exists(path):
            res[tag] = "golden file missing"
            continue
        g = np.load(path, allow_pickle=False)
        B0 = int(g["B"])
        if batch["input_ids"].shape[0] < B0 or not np.array_equal(batch["input_ids"][:B0].cpu().numpy(), g["in::input_ids"]):
            res[tag] = "bench batch differs from the golden's batch"
            continue
        D.load_seeded_weights(model, lit, seed=int(g["weight_seed"]), conditioned=cond)
        ref, ctl = g["mask_logits"], g["ctl::mask_logits"]
        lab = np.asarray(g["in::label"])
        rk = lambda x: (x > x[np.arange(B0), lab][:, None]).sum(1) + 1
        order = lambda x: np.argsort(-x, axis=1, kind="stable")

        def against_reference(lg):
            return {"max_abs_dlogit": round(float(np.abs(lg - ref).max()), 6), "rms_dlogit": round(float(np.sqrt(((lg - ref) ** 2).mean())), 7),
                    "label_ranks_identical_to_reference": f"{int((rk(lg) == rk(ref)).sum())}/{B0}",
                    "top10_entity_indices_identical": f"{int((order(lg)[:, :10] == order(ref)[:, :10]).all(1).sum())}/{B0}"}
        row = {"logit_abs_max": round(float(np.abs(ref).max()), 3), "n_logits": int(ref.size),
               "timed_weights": bool((timed_weights == "conditioned") == cond and timed_weights in ("conditioned", "g7plain"))}
        # bf16 twice: with the forward pass of the TIMED training step (LayerNorm as its own pass; also what a no_grad pass runs by default), and with the
        # opt-in fold of the vision LayerNorms into Q/K/V and fc1 (engine.ln_fold: no_grad passes only)
        fold0 = getattr(model.engine, "ln_fold", False)
        for prec, fold, key in (("bf16", False, "bf16_timed_path"), ("bf16", True, "bf16_no_grad_path_with_ln_fold_opt_in"), ("fp32", fold0, "fp32_path_evaluation_default")):
            if fold and not hasattr(model.engine, "ln_fold"):
                continue
            model.set_precision(prec)
            if hasattr(model.engine, "ln_fold"):
                model.engine.ln_fold = fold
            try:
                with torch.no_grad():
                    pos = (batch["input_ids"][:B0] == D.MASK).int().argmax(1)
                    kw = dict(needed_rows=pos) if (prec == "bf16" and lit.last_layer_rows) else {}
                    o, _ = model(**{k: batch[k][:B0] for k in keys}, return_dict=True, **kw)
                    lg = o.logits.mask_rows(batch["input_ids"][:B0], D.MASK)[:, ids].float().cpu().numpy()
            finally:
                model.set_precision("bf16")
                if hasattr(model.engine, "ln_fold"):
                    model.engine.ln_fold = fold0
            r = against_reference(lg)
            r["meets_north_star_logit_tol"] = bool(r["max_abs_dlogit"] < (1e-2 if prec == "bf16" else 1e-3))
            row[key] = r
        row["reference_bf16_weight_control"] = {"max_abs_dlogit": round(float(np.abs(ctl - ref).max()), 5),
                                                "rms_dlogit": round(float(np.sqrt(((ctl - ref) ** 2).mean())), 6),
                                                "label_ranks_identical_to_reference": f"{int((rk(ctl) == rk(ref)).sum())}/{B0}",
                                                "what": "the reference itself, fp32 math, with only its weight matrices rounded to bf16"}
        res[tag] = row
    return res


def self_launch(n: int) -> int:
    """Re-execute this command line under ``python -m torch.distributed.run`` with ``n`` ranks on this node (one process per GPU, rendezvous on
    127.0.0.1, a free port), exactly the command the driver uses for N > 1; stdout / stderr pass through, so rank 0's JSON line is this process's."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


class PowerSampler:
    """Package power and shader clock of GPU ``index`` sampled from a helper thread while a timed region runs (rocm-smi, a few Hz; a separate
    process per sample, so the launching thread is not slowed).  ``median()`` -> (W, MHz, n) or (None, None, 0) where the tool is absent."""

    def __init__(self, index: int = 0):
        import shutil
        self.index, self.samples, self._stop, self._th = index, [], False, None
        self.tool = shutil.which("rocm-smi") or ("/opt/rocm/bin/rocm-smi" if os.path.exists("/opt/rocm/bin/rocm-smi") else None)

    def _run(self):
        import re
        import subprocess
        while not self._stop:
            try:
                out = subprocess.run([self.tool, "-d", str(self.index), "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
            except Exception:
                return
            p = re.search(r"Power \(W\): ([0-9.]+)", out)
            c = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out) or re.search(r"sclk clock level: \S+: \((\d+)Mhz\)", out)
            if p and c:
                self.samples.append((time.perf_counter(), float(p.group(1)), int(c.group(1))))

    def __enter__(self):
        if self.tool:
            import threading
            self._th = threading.Thread(target=self._run, daemon=True)
            self._th.start()
        self.t0 = time.perf_counter()
        return self

    def __exit__(self, *a):
        self.t1 = time.perf_counter()
        self._stop = True

    def median(self):
        if self._th is not None:
            self._th.join(timeout=15)
        xs = [(w, c) for t, w, c in self.samples if self.t0 <= t <= self.t1 + 0.05]
        if not xs:
            return None, None, 0
        med = lambda v: sorted(v)[len(v) // 2]
        return med([w for w, _ in xs]), med([c for _, c in xs]), len(xs)


def device_allocs() -> int:
    """hipMalloc calls the caching allocator has made so far: a timed region that grows the pool (a block still held by a side queue's
    record_stream when the next step asks for it) pays for the allocation -- the count over the region is reported next to its time."""
    return int(torch.cuda.memory_stats().get("num_device_alloc", 0))


def timed_leg(tr, lit, batch, n_warm: int, n_steps: int, barrier, first_idx: int = 0):
    """``n_warm`` untimed + ``n_steps`` timed training steps of an already set-up trainer; seconds of the timed part
    (``timed_leg.allocs``: device allocations inside it)."""
    seq = batch if isinstance(batch, (list, tuple)) else [batch]      # (a list: one batch after the other, round robin)
    for i in range(n_warm):
        tr.train_step(lit, seq[i % len(seq)], first_idx + i)
    tr.settle_pool()                                                  # (fewer than three warm-up steps: the pool headroom now, not inside the timed steps)
    barrier()
    n0 = device_allocs()
    t1 = time.perf_counter()
    for i in range(n_steps):
        tr.train_step(lit, seq[(n_warm + i) % len(seq)], first_idx + n_warm + i)
    barrier()
    dt = time.perf_counter() - t1
    timed_leg.allocs = device_allocs() - n0
    return dt


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
    ap.add_argument("--entity-head", type=int, default=11292, choices=[2063, 11292],
                    help="fine-tune scoring head: 11292 = every MarKG entity (north_star's '~11k-entity head', the headline), 2063 = the MARS "
                         "analogy entities the reference's fine-tune branch scores (lit_models/transformer.py:95); the other one is timed "
                         "briefly as well and reported under 'alt_entity_head'")
    ap.add_argument("--weights", default=None, choices=["plain", "conditioned", "g7plain"],
                    help="weights of the TIMED network: conditioned (default for the MKGformer fine-tune step) / g7plain = the seeded weight sets of the reference "
                         "goldens tests/golden/g7_bench_{cond,plain}.npz -- the network that is timed is the network the `parity` block compares with the UNMODIFIED "
                         "reference (first 32 examples of this very batch); plain = torch-RNG N(0,0.02) (rounds 1-4's headline; timed briefly as `alt_weights`). "
                         "The step time does not depend on the values (r04: 2892 vs 2896 examples/s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=3, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-rate-only", type=int, default=0, help=argparse.SUPPRESS)      # child process of cpu_baseline(): threads
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--train-only", action="store_true", help="skip the evaluation pass (Hits@1), the precision comparison and the alternate-head run: "
                                                              "the process then launches nothing but training steps (rocprofv3 traces whose per-kernel averages are per training step)")
    a = ap.parse_args()
    if a.cpu_rate_only:
        print(cpu_step_rate(a.patch, a.seq_len, a.cpu_rate_only, a.cpu_iters), flush=True)
        return

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(self_launch(a.gpus))                # `python bench.py --gpus N` as typed: spawn the N ranks (one per GPU) and relay rank 0's line

    from mkg_analogy_amd import data_synth as D
    from mkg_analogy_amd import ops
    from mkg_analogy_amd.distributed import init_from_env
    from mkg_analogy_amd.trainer import Trainer
    import torch.distributed as dist

    rank, local, world = init_from_env()
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}, "
                         f"or run `python bench.py --gpus {a.gpus}` without WORLD_SIZE / RANK in the environment (it then starts its own ranks)")
    ops.require_gpu()
    dev = torch.device("cuda", local)
    pre = a.task == "pretrain"
    if a.weights is None:
        a.weights = "conditioned" if (a.model == "mkgformer" and not pre) else "plain"
    head = D.N_ENT if pre else a.entity_head
    model, lit, cfg = build(a.patch, seed=0, device=dev, backbone=a.model, entity_head=D.N_ANALOGY if pre else head)
    if a.weights != "plain":
        assert a.model == "mkgformer", "--weights conditioned / g7plain are MKGformer weight sets"
        D.load_seeded_weights(model, lit, seed=0, conditioned=a.weights == "conditioned")
    if pre:
        lit.args.pretrain = 1
    batch = D.make_batch(a.batch, a.seq_len, seed=1234 + rank, device=dev, pretrain=pre, n_labels=head)
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
    tr.settle_pool()                                 # (--warmup < 3: the allocator's pool headroom now, not inside the timed steps)
    barrier()
    allocs0 = device_allocs()
    with PowerSampler(local) as psamp:               # (helper thread; rank 0 reports it)
        t0 = time.perf_counter()
        for i in range(a.steps):
            loss = tr.train_step(lit, batch, a.warmup + i)
        barrier()
        dt = time.perf_counter() - t0
    allocs_timed = device_allocs() - allocs0
    power_w, sclk_mhz, n_power = psamp.median() if rank == 0 else (None, None, 0)
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    ms = 1000.0 * dt / a.steps
    P = (224 // a.patch) ** 2
    value = a.batch * world * a.steps / dt
    train_gflop = 3.0 * (fwd_gflop_per_example(P, a.seq_len, head) if a.model == "mkgformer" else flava_fwd_gflop_per_example(P, a.seq_len, head))

    roof = None
    if not a.no_kernel_timing:
        # The timed region above runs the weight-gradient GEMMs (gemm_tn) and the text layers on side streams, so a launch shares
        # the CUs with other kernels part of the time and its start-to-end time stops being the kernel's own rate.  The kernels
        # are therefore timed in one extra step with both overlaps switched off (MART_OVERLAP_WGRAD=0 MART_TWO_STREAM=0 gives the
        # same schedule for rocprofv3: profiles/*_serial*); the overlapped figure of the dominant kernel is reported next to it.
        eng = model.engine
        ov, ts = getattr(eng, "overlap_wgrad", False), getattr(eng, "two_stream", False)
        with KernelTimer(ops) as kt_ov:
            tr.train_step(lit, batch, a.warmup + a.steps)
        fam_ov = kt_ov.summary()
        eng.overlap_wgrad = False
        eng.two_stream = False
        with KernelTimer(ops) as kt:
            tr.train_step(lit, batch, a.warmup + a.steps + 1)
        eng.overlap_wgrad, eng.two_stream = ov, ts
        fam = kt.summary()
        n, fl, kms, by = fam.get("gemm_nt_256", [0, 0.0, 0.0, 0.0])
        n_ov, fl_ov, kms_ov, _ = fam_ov.get("gemm_nt_256", [0, 0.0, 0.0, 0.0])
        ach = fl / (kms * 1e-3) / 1e12 if kms > 0 else 0.0
        # HBM traffic of the dominant kernel: PMC passes of this same command (tools/pmc.sh -> profiles/rNN_pmc_gemm_nt.json, newest round first).
        # The file names the kernel source it was collected on; when gemm_nt.hip has changed since, the number is withheld instead of going stale.
        traffic, traffic_note = None, "no PMC collection for this configuration"
        import glob
        pmcs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_gemm_nt.json")), reverse=True)
        if pmcs and a.batch == 256 and a.patch == 16 and a.seq_len == 64 and a.model == "mkgformer" and not pre:
            sha = source_sha16("gemm_nt.hip", "common.h")
            traffic_note = f"profiles/{os.path.basename(pmcs[0])} was collected on an older gemm_nt.hip: withheld"
            for pmc in pmcs:
                pj = json.load(open(pmc))
                if pj.get("source_sha16") == sha:
                    traffic = pj.get("hbm_bytes_per_launch")
                    traffic_note = f"rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, {pj.get('launches_averaged')} launches (profiles/{os.path.basename(pmc)})"
                    break

        def famrow(key, name, bound_tf=2500.0):
            c, f, m, _ = fam.get(key, [0, 0.0, 0.0, 0.0])
            return None if not c or m <= 0 else {"kernel": name, "launches_per_step": c, "ms_per_step": round(m, 3), "achieved": round(f / (m * 1e-3) / 1e12, 1),
                                                 "unit": "TFLOP/s", "frac": round(f / (m * 1e-3) / 1e12 / bound_tf, 4)}
        def hbmrow(key, name):
            c, _, m, by = fam.get(key, [0, 0.0, 0.0, 0.0])
            return None if not c or m <= 0 else {"kernel": name, "bound": "hbm", "launches_per_step": c, "ms_per_step": round(m, 3), "achieved": round(by / (m * 1e-3) / 1e12, 2),
                                                 "peak": 8.0, "unit": "TB/s", "frac": round(by / (m * 1e-3) / 8e12, 4), "algorithmic_mb_per_launch": round(by / c / 1e6, 1)}
        others = [r for r in (famrow("gemm_tn", "gemm_tn8_kernel + tn_reduce_k (weight / bias gradients, deterministic split reduction)"),
                              famrow("attn_fwd_vision", "attn_fwd_k<vision> (393 queries x 393 / 457 keys, d 64)"),
                              famrow("attn_bwd_vision", "attn_bwd_fused_k <vision> (one workgroup per head: dQ, dK, dV in one pass; algorithmic flops = 2 x forward)"),
                              famrow("gemm_nt_128", "gemm_nt_kernel<128,128,2,2> (small products: head, short grids)"),
                              hbmrow("ln_bwd_vision", "ln_bwd_fast_k (vision stream: dy bf16 + x f32 + residual gradient bf16 in, bf16 total out: 10 bytes per element; 16 with MART_GRAD_STREAM_BF16=0)"),
                              hbmrow("ln_fwd_vision", "ln_fwd_fast_k (vision stream: x f32 in, bf16 out)")) if r]
        roof = {"bound": "mfma", "kernel": "gemm_nt_kernel<256,256,2,4> (bf16 MFMA 16x16x32 NT GEMM, 4-phase K loop, fused epilogues)", "achieved": round(ach, 1),
                "peak": 2500.0, "unit": "TFLOP/s", "frac": round(ach / 2500.0, 4), "traffic": traffic, "traffic_unit": "HBM bytes per launch", "traffic_source": traffic_note,
                "algorithmic_bytes_per_launch": round(by / max(n, 1)), "launches_per_step": n, "ms_per_step": round(kms, 3),
                "avg_launch_ms": round(kms / max(n, 1), 4), "algorithmic_gflop_per_launch": round(fl / max(n, 1) / 1e9, 1),
                "step_frac_of_mfma_peak": round(value / world * train_gflop / 2.5e6, 4),
                # the roofline at the clock the box actually granted during the timed region (the chip clocks to its package power cap: 2.5 PF is the
                # 2.4 GHz figure); null when rocm-smi could not be sampled
                "power_w": power_w, "sclk_mhz": sclk_mhz, "power_samples": n_power,
                "frac_at_granted_clock": round(ach / (2500.0 * sclk_mhz / 2400.0), 4) if sclk_mhz else None,
                "step_frac_at_granted_clock": round(value / world * train_gflop / 2.5e6 / (sclk_mhz / 2400.0), 4) if sclk_mhz else None,
                "power_note": "medians of rocm-smi --showpower --showclocks sampled by a helper thread during the timed region of the headline steps",
                "timing": "HIP events around every launch, one step with the side streams (weight gradients, text layers) off: kernel alone on the GPU",
                "achieved_with_wgrad_overlap": round(fl_ov / (kms_ov * 1e-3) / 1e12, 1) if kms_ov > 0 else None,
                "avg_launch_ms_with_wgrad_overlap": round(kms_ov / max(n_ov, 1), 4), "other_kernels": others}
    comm = None
    if world > 1 and getattr(tr.sync, "reducer", None) is not None and not a.train_only:
        # Communication diagnostics (outside the timed region: the statistics need one host sync per step): per-bucket start / end events
        # on the collective stream against the end of the backward pass on the compute stream
        red = tr.sync.reducer
        red.timing = True
        red.stats(reset=True)
        for i in range(3):
            tr.train_step(lit, batch, a.warmup + a.steps + 2 + i)
        barrier()
        st_ = red.stats(reset=True)
        red.timing = False
        comm = {"rccl_world": dist.get_world_size(), "backend": dist.get_backend(), "buckets": st_["buckets"], "bucket_mb": st_["bucket_mb"],
                "bucket_dtype": st_["bucket_dtype"], "comm_ms_per_step": round(st_["comm_ms"], 3), "comm_exposed_ms": round(st_["comm_exposed_ms"], 3),
                "steps_measured": st_["steps"],
                "what": "all-reduce time per step (sum over buckets, collective stream) and the part still running after the backward pass had ended"}
    # Hits@1 of the (untrained, random-init) model on the same batch -- reported to exercise the ranking eval path (fp32-accurate pass: the
    # evaluation default, so the ranks behind this number are the reference's wherever its own margins are not rounding-level)
    metrics = tr.validate(lit, [batch]) if not a.train_only else {}
    held_out_hits1 = None
    if not a.train_only and not pre:
        # the same pass over a batch the network has NOT been stepped on (random labels: chance level, 1 / entity_head)
        hb = D.make_batch(a.batch, a.seq_len, seed=99991 + rank, device=dev, pretrain=pre, n_labels=head)
        held_out_hits1 = tr.validate(lit, [hb]).get("Eval_entity/hits1")
        del hb
    evalb = None
    if world == 1 and not a.train_only and not a.no_kernel_timing:
        # Evaluation-path throughput (lit_models/transformer.py:115-166: forward + scoring + rank of the label), per precision mode
        def time_eval(prec, n=3):
            lit.args.eval_precision = prec
            try:
                m = tr.validate(lit, [batch])
                barrier()
                t1 = time.perf_counter()
                for _ in range(n):
                    m = tr.validate(lit, [batch])
                barrier()
                return {"examples_per_s": round(a.batch * n / (time.perf_counter() - t1), 1), "hits1": m.get("Eval_entity/hits1"),
                        "mean_rank": m.get("Eval_entity/mean_rank")}
            finally:
                lit.args.eval_precision = None
        evalb = {"what": "validation pass over the timed batch: forward, scoring head, device-side rank of the label",
                 "bf16": time_eval("bf16"),                              # the training configuration (text stream on fp16 operands, split-precision head)
                 "fp32": time_eval("fp32")}
        if hasattr(model.engine, "ln_fold") and not model.engine.ln_fold:   # the bf16 pass with the vision LayerNorms folded into Q/K/V and fc1 (opt-in, MART_LN_FOLD=1)
            model.engine.ln_fold = True
            try:
                evalb["bf16_ln_fold"] = time_eval("bf16")
            finally:
                model.engine.ln_fold = False
        evalb["eval_examples_per_s"] = evalb["fp32"]["examples_per_s"]      # the default of validation / test passes (TransformerLitModel._eval_at)
        evalb["eval_precision"] = "fp32"
    tsplit = None
    if world == 1 and a.model == "mkgformer" and not a.no_kernel_timing and not a.train_only:
        # the training step with the text stream in the other precision mode (plain bf16 operands), timed briefly on the same network
        eng_ = model.engine
        eng_.text_f16 = not eng_.text_f16
        for i in range(2):
            tr.train_step(lit, batch, a.warmup + a.steps + 20 + i)
        barrier()
        t1 = time.perf_counter()
        for i in range(5):
            tr.train_step(lit, batch, a.warmup + a.steps + 22 + i)
        barrier()
        d3 = time.perf_counter() - t1
        tsplit = {"text_f16": eng_.text_f16, "steps": 5, "ms_per_step": round(1000.0 * d3 / 5, 3), "value": round(a.batch * 5 / d3, 2)}
        eng_.text_f16 = not eng_.text_f16
    alt = None
    if world == 1 and a.model == "mkgformer" and not pre and not a.no_kernel_timing and not a.train_only:
        # the other scoring head, timed briefly on the same network (the head is 0.03 % of the step's FLOPs either way)
        other = D.N_ANALOGY if head == D.N_ENT else D.N_ENT
        lit.analogy_entity_ids = D.data_config()["analogy_entity_ids"] if other == D.N_ANALOGY else list(range(D.BASE_VOCAB, D.BASE_VOCAB + D.N_ENT))
        lit._ids_cache.clear()
        b2 = dict(batch)
        b2["label"] = batch["label"] % other
        for i in range(2):
            tr.train_step(lit, b2, a.warmup + a.steps + 2 + i)
        barrier()
        t1 = time.perf_counter()
        for i in range(5):
            tr.train_step(lit, b2, a.warmup + a.steps + 4 + i)
        barrier()
        d2 = time.perf_counter() - t1
        alt = {"entity_head": other, "steps": 5, "ms_per_step": round(1000.0 * d2 / 5, 3), "value": round(a.batch * 5 / d2, 2)}
        # back to the headline's head: the legs below train on `batch`, whose labels index THAT head (a label beyond the 2063-entity head is out of range:
        # NaN loss since round 6, an out-of-bounds read before -- rounds 2-5 timed `alt_weights` that way)
        lit.analogy_entity_ids = D.data_config()["analogy_entity_ids"] if head == D.N_ANALOGY else list(range(D.BASE_VOCAB, D.BASE_VOCAB + D.N_ENT))
        lit._ids_cache.clear()
    altw = None
    if world == 1 and a.model == "mkgformer" and not pre and a.weights == "conditioned" and not a.no_kernel_timing and not a.train_only:
        # the same step on PLAIN N(0, 0.02) weights (the golden G7-plain set), timed briefly: the rate does not depend on the values
        D.load_seeded_weights(model, lit, seed=0, conditioned=False)
        for i in range(2):
            tr.train_step(lit, batch, a.warmup + a.steps + 40 + i)
        barrier()
        t1 = time.perf_counter()
        for i in range(5):
            tr.train_step(lit, batch, a.warmup + a.steps + 42 + i)
        barrier()
        d4 = time.perf_counter() - t1
        altw = {"weights": "g7plain (plain N(0,0.02), seeded)", "steps": 5, "ms_per_step": round(1000.0 * d4 / 5, 3), "value": round(a.batch * 5 / d4, 2)}
    parity = None
    if world == 1 and a.model == "mkgformer" and not pre and a.patch == 16 and a.seq_len == 64 and not a.no_kernel_timing and not a.train_only:
        parity = reference_parity(model, lit, batch, cfg, dev, a.weights)
    altg = None
    if world == 1 and a.model == "mkgformer" and not pre and a.patch == 16 and not a.no_kernel_timing and not a.train_only:
        # The reference's own default geometry (/root/reference/MarT/main.py:79: CLIP ViT-B/32, 49 patches per image, 99 vision tokens): the one
        # where north_star's 10 k examples/s lies below the MFMA roofline.  Same step, same batch size, a second network; 4 warm-up + 5 timed steps.
        m2, lit2, _ = build(32, seed=0, device=dev, backbone=a.model, entity_head=head)
        D.load_seeded_weights(m2, lit2, seed=0, conditioned=True)
        tr2 = Trainer(max_epochs=1, max_steps=200, world_size=1)
        tr2._setup(lit2, [None] * 200)
        d5 = timed_leg(tr2, lit2, batch, 4, 5, barrier)
        v5 = a.batch * 5 / d5
        gf5 = 3.0 * fwd_gflop_per_example(49, a.seq_len, head)
        altg = {"what": "the reference's default geometry (MarT/main.py:79, CLIP ViT-B/32): 49 patches per image, 99 vision tokens; everything else as the headline",
                "patches_per_image": 49, "vision_tokens": 99, "steps": 5, "warmup": 4, "ms_per_step": round(1000.0 * d5 / 5, 3), "value": round(v5, 2),
                "train_gflop_per_example": round(gf5, 1), "step_frac_of_mfma_peak": round(v5 * gf5 / 2.5e6, 4),
                "roofline_examples_per_s": round(2.5e6 / gf5, 1), "device_allocs_in_timed_region": timed_leg.allocs}
        # ... and the same network on batches padded as the reference's collate pads them (data_module.py:113-119: to the longest example of the batch;
        # MARS examples are 40 .. 57 tokens): the length changes from step to step, no kernel or buffer is specialised on it (tools/var_len.py:
        # 60-step runs, profiles/r06_var_len.txt).
        import random
        rng_ = random.Random(0)
        lens_ = [rng_.randint(40, 57) for _ in range(14)]
        seq_ = [D.make_batch(a.batch, L_, seed=4321 + i, device=dev, pretrain=False, n_labels=head) for i, L_ in enumerate(lens_)]
        d10 = timed_leg(tr2, lit2, seq_, 4, 10, barrier, first_idx=9)
        altg["real_lengths"] = {"what": "padded length drawn from 40 .. 57 per batch (the reference pads a batch to its longest example)", "seq_lens_timed": lens_[4:],
                                "steps": 10, "warmup": 4, "ms_per_step": round(1000.0 * d10 / 10, 3), "value": round(a.batch * 10 / d10, 2), "unit": "examples/s",
                                "device_allocs_in_timed_region": timed_leg.allocs}
        del seq_
    altt = altm = alts = None
    if world == 1 and a.model == "mkgformer" and not pre and a.patch == 16 and not a.no_kernel_timing and not a.train_only:
        # BASELINE configs[4] and configs[3] on the driver's box, briefly (4 warm-up + 5 timed steps each: the second in-flight step's buffers are allocated in step 2, the pool headroom after step 3): the MarKG pre-train step (L = 96, LSCE over the
        # full 11 292-entity / 192-relation slices, pre_type 50 / 50, no sep_idx) and the FLAVA backbone (12 + 12 + 6 layers, 393 image tokens, B = 256).
        # The full-length runs with per-kernel tables are profiles/r06_bench_{pretrain,flava}.json (python bench.py --task pretrain --seq-len 96 / --model flava).
        def leg(backbone, task_pre, L_, what):
            try:
                torch.cuda.empty_cache()
                m_, lit_, _ = build(a.patch, seed=0, device=dev, backbone=backbone, entity_head=D.N_ANALOGY if task_pre else head)
                if task_pre:
                    lit_.args.pretrain = 1
                b_ = D.make_batch(a.batch, L_, seed=1234, device=dev, pretrain=task_pre, n_labels=D.N_ENT if task_pre else head)
                tr_ = Trainer(max_epochs=1, max_steps=200, world_size=1)
                tr_._setup(lit_, [None] * 200)
                d_ = timed_leg(tr_, lit_, b_, 4, 5, barrier)
                v_ = a.batch * 5 / d_
                gf_ = 3.0 * (fwd_gflop_per_example(P, L_, D.N_ENT if task_pre else head) if backbone == "mkgformer" else flava_fwd_gflop_per_example(P, L_, head))
                return {"what": what, "batch": a.batch, "seq_len": L_, "patches_per_image": P, "steps": 5, "warmup": 4, "ms_per_step": round(1000.0 * d_ / 5, 3),
                        "value": round(v_, 2), "unit": "examples/s", "train_gflop_per_example": round(gf_, 1), "step_frac_of_mfma_peak": round(v_ * gf_ / 2.5e6, 4),
                        "device_allocs_in_timed_region": timed_leg.allocs}
            except Exception as e:                      # a failure of a side leg must not lose the headline measurement
                return {"what": what, "error": f"{type(e).__name__}: {e}"[:300]}
            finally:
                torch.cuda.empty_cache()
        m2 = lit2 = tr2 = None                          # (the 49-patch network of alt_geometry)
        alts = leg("mkgformer", False, 57, "a length real MARS batches have: L = 57 (the reference pads a batch to its longest example, data_module.py:113-119; "
                   "real lengths 40 .. 57), everything else as the headline -- no kernel of the step needs L to be a multiple of 32 / 64")
        if "value" in alts:
            alts["tokens_per_s_vs_headline"] = round(alts["value"] * (1 + 2 * P + 57) / (value * (1 + 2 * P + a.seq_len)), 4)
            alts["text_tokens_per_s_vs_headline"] = round(alts["value"] * 57 / (value * a.seq_len), 4)
        altt = leg("mkgformer", True, 96, "BASELINE configs[4]: MarKG pre-train step (link prediction), L = 96, heads E = 11292 / R = 192, pre_type 50/50, no sep_idx")
        altm = leg("flava", False, a.seq_len, "BASELINE configs[3]: FLAVA-base backbone (12 image + 12 text + 6 multimodal layers, 393 image tokens), fine-tune step")
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
               "vs_baseline": None, "device_allocs_in_timed_region": allocs_timed,
               "dtype": "bf16" + (" (text-stream forward operands fp16, f32 accumulate)" if getattr(getattr(model, "engine", None), "text_f16", False) else ""),
               "data": "synthetic",
               "config": {"workload": (f"MKGformer (BERT-base + ViT-B/{a.patch} patches)" if a.model == "mkgformer" else "FLAVA-base (12+12+6 layers)") +
                          (" fine-tune step, MARS-shaped batch" if not pre else " MarKG pre-train step (full entity / relation heads)"), "batch_per_gpu": a.batch,
                          "global_batch": a.batch * world, "seq_len": a.seq_len, "patches_per_image": P, "vision_tokens": 1 + 2 * P,
                          "entity_head": head, "relation_head": D.N_REL if pre else None, "vocab": D.VOCAB, "parallelism": f"dp{world}",
                          "weights": {"plain": "random-init N(0,0.02) (torch RNG)", "g7plain": "random-init N(0,0.02), the seeded set of golden G7 (plain)",
                                      "conditioned": "random-init N(0,0.02), the seeded, well-conditioned set of golden G7 (text value projections of layers 8-11 x 0.05)"}[a.weights]},
               "loss": round(float(loss), 4),
               # NOT an accuracy: the ranking eval of the timed batch after the network has taken steps + warmup (+ diagnostics) updates ON that very
               # batch (memorised: -> 1.0).  No trained weights / real MARS split exist offline; the line exercises the Hits@1 path, nothing more.
               "hits1_on_the_timed_batch": metrics.get("Eval_entity/hits1"), "hits1_held_out_synthetic_batch": held_out_hits1,
               "train_gflop_per_example": round(train_gflop, 1)}
        if spread is not None:
            out["replica_param_checksum_spread"] = spread        # 0.0: all ranks hold identical weights
        if roof is not None:
            out["roofline"] = roof
        if parity is not None:
            out["parity"] = parity
        if alt is not None:
            out["alt_entity_head"] = alt
        if altw is not None:
            out["alt_weights"] = altw
        if altg is not None:
            out["alt_geometry"] = altg
        if alts is not None:
            out["alt_seq_len"] = alts
        if altt is not None:
            out["alt_task"] = altt
        if altm is not None:
            out["alt_model"] = altm
        if tsplit is not None:
            out["alt_text_precision"] = tsplit
        if evalb is not None:
            out["eval"] = evalb
            out["eval_examples_per_s"] = evalb["eval_examples_per_s"]
            out["eval_precision"] = evalb["eval_precision"]
        if comm is not None:
            out["comm"] = comm
            out["comm_exposed_ms"] = comm["comm_exposed_ms"]
            out["rccl_world"] = comm["rccl_world"]
        try:                                            # bad labels / examples without [MASK] flagged by any kernel of this process (functional.check_status)
            from mkg_analogy_amd import functional as Fn_
            Fn_.check_status()
            out["status_check"] = "ok"
        except IndexError as e:
            out["status_check"] = f"FLAGGED: {e}"
        out["metric_detail"] = f"entity_head={head}, text_f16={int(getattr(getattr(model, 'engine', None), 'text_f16', False))}, weights={a.weights}"
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
