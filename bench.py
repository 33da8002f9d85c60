#!/usr/bin/env python3
"""Headline benchmark: train spectra/s on synthetic spectra of the local/real_config shape (BASELINE.json
configs[1]/[2]: E1600 H16 L8 stablemax, per-GPU batch 32, bf16 compute with f32 master weights), one process per
GPU, gradient all-reduce on RCCL.  Also reports, in the same JSON line: the roofline of the fused dwconv7+LN kernel
(K1) measured live with HIP events, greedy SMILES tokens/s with the KV cache (nocat shape), and the CPU baseline
(the oracle's train step timed on the host cores, rank 0 at N=1 only).

    python bench.py [--gpus N --steps K --warmup W]          # N>1: launched by torch.distributed.run
"""
import argparse
import json
import os
import sys
import time

if int(os.environ.get("RANK", "0")) == 0:       # the first import of torch on a fresh box can take minutes
    print(f"[bench {time.strftime('%H:%M:%S')}] importing torch", file=sys.stderr, flush=True)
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # configs/local_config.yaml == configs/real_config.yaml (model section)
    "local_config": dict(E=1600, H=16, L=8, sm=True, max_seq=95, max_nmr=190, max_mem=32),
    "local_config_nocat": dict(E=1024, H=8, L=4, sm=False, max_seq=80, max_nmr=128, max_mem=64),
    "test_config": dict(E=768, H=4, L=8, sm=True, max_seq=95, max_nmr=190, max_mem=32),
    "real_config_mup": dict(E=2048, H=4, L=8, sm=True, max_seq=95, max_nmr=190, max_mem=32),
}
K1_PMC = os.path.join("profiles", "r03_k1_pmc.json")
K4_PMC = os.path.join("profiles", "r03_k4_pmc.json")
ATTN_PMC = os.path.join("profiles", "r03_attn_d100_pmc.json")
ATTN_KERNELS = "attn_fwd_r_k / attn_bwd_dq_r_k + attn_bwd_dkv_r_k (whole head resident in LDS; attention_res.hip)"
ATTN_KERNELS_WIDE = "attn_fwd_w_k / attn_delta_k + attn_bwd_dq_w_k + attn_bwd_dkv_w_k (head_dim 512; attention_wide.hip)"
HBM_PEAK_GBPS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA peak (same guide)


def log(msg):
    """Progress on stderr (stdout carries exactly one JSON line): a silent minute-long phase looks like a hang."""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def build_model(c, device, seed=0):
    from multimodal_amd.models import MultiModalToSMILESModel
    torch.manual_seed(seed)
    m = MultiModalToSMILESModel(591, 1000, max_seq_length=c["max_seq"], max_nmr_length=c["max_nmr"],
                                max_memory_length=c["max_mem"], embed_dim=c["E"], num_heads=c["H"], num_layers=c["L"],
                                dropout=0.0, verbose=False, use_stablemax=c["sm"])
    return m.to(device)


def synth_batch(c, B, device, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    ir = torch.rand(B, 1000, generator=g)
    nmr = torch.randint(0, 1000, (B, c["max_nmr"]), generator=g)
    tgt = torch.randint(15, 591, (B, c["max_seq"]), generator=g)
    tgt[:, 0] = 12
    return ir.to(device), nmr.to(device), tgt.to(device)


def k1_roofline(device, dtype=torch.float32, B=8192, C=64, Ln=250, iters=20):
    """Algorithmic bytes of K1 forward = B*C*L*(s_in+s_out) over the average launch time measured
    with HIP events on the launch stream."""
    from multimodal_amd import ops
    x = torch.rand(B, Ln, C, device=device).to(dtype)
    w = torch.randn(C, 1, 7, device=device) * 0.3
    b = torch.zeros(C, device=device)
    lw = torch.ones(C, device=device)
    lb = torch.zeros(C, device=device)
    for _ in range(3):
        ops.dwconv7_ln_fwd(x, w, b, lw, lb)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        ops.dwconv7_ln_fwd(x, w, b, lw, lb)
    e.record()
    torch.cuda.synchronize()
    t = s.elapsed_time(e) / iters * 1e-3
    nbytes = B * C * Ln * 2 * x.element_size()
    ach = nbytes / t / 1e9
    # HBM bytes per launch from the PMC passes (cannot be collected live: rocprofv3 --pmc runs benchmarks/k1_pmc.py on
    # this same shape; benchmarks/pmc_summary.py applies the gfx950 FETCH_SIZE x2 correction) -- fp32 shape only
    traffic = None
    pmc = os.path.join(ROOT, K1_PMC)
    if dtype == torch.float32 and (B, C, Ln) == (8192, 64, 250) and os.path.exists(pmc):
        with open(pmc) as fh:
            traffic = int(json.load(fh)["traffic_bytes_per_launch"])
    return {"kernel": "dwconv7_ln_fwd_strip", "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": traffic, "launch_us": round(t * 1e6, 2),
            "traffic_source": (K1_PMC + " (rocprofv3 --pmc passes over benchmarks/k1_pmc.py, same shape; not re-measured "
                               "in this run)") if traffic is not None else None,
            "shape": {"B": B, "C": C, "L": Ln, "dtype": str(dtype).replace("torch.", "")},
            "algorithmic_bytes_per_launch": nbytes}


def k1_all_shapes(device):
    """K1 / K1b on the four encoder stage shapes (C, L) x {f32, bf16} x {fwd, bwd} at a saturating batch (traffic >= 0.4 GB
    per launch): algorithmic bytes B*L*C*(s_in+s_out) forward, B*L*C*(2*s_in+s_out) backward, HIP events."""
    from multimodal_amd import ops
    out = []
    for dtype in (torch.float32, torch.bfloat16):
        for (B, C, Ln) in [(8192, 64, 250), (8192, 128, 125), (8192, 256, 62), (2048, 1600, 31), (2048, 2048, 31)]:
            x = torch.rand(B, Ln, C, device=device).to(dtype)
            w = torch.randn(C, 1, 7, device=device) * 0.3
            b = torch.zeros(C, device=device); lw = torch.ones(C, device=device); lb = torch.zeros(C, device=device)
            _, mean, rstd = ops.dwconv7_ln_fwd(x, w, b, lw, lb, save_stats=True)
            dy = torch.randn_like(x)
            tf = _event_time(lambda: ops.dwconv7_ln_fwd(x, w, b, lw, lb), iters=10)
            tb = _event_time(lambda: ops.dwconv7_ln_bwd(x, dy, None, w, b, lw, mean, rstd), iters=10)
            nb = B * C * Ln * x.element_size()
            out.append({"C": C, "L": Ln, "B": B, "dtype": str(dtype).replace("torch.", ""),
                        "fwd_us": round(tf * 1e6, 1), "fwd_GBps": round(2 * nb / tf / 1e9), "fwd_frac": round(2 * nb / tf / 1e9 / HBM_PEAK_GBPS, 3),
                        "bwd_us": round(tb * 1e6, 1), "bwd_GBps": round(3 * nb / tb / 1e9), "bwd_frac": round(3 * nb / tb / 1e9 / HBM_PEAK_GBPS, 3)})
            del x, dy
    return out


def gemm_roofline(device, M=10080, N=1600, K=6400, iters=20):
    """The decoder's d_ffn -> d_model projection on the large-tile MFMA kernel (bias + residual fused): 2*M*N*K flops
    over the average launch time (HIP events on the launch stream) against the dense bf16 MFMA peak."""
    from multimodal_amd import ops
    a = (torch.rand(M, K, device=device) * 2 - 1).to(torch.bfloat16)
    w = (torch.rand(N, K, device=device) * 2 - 1).to(torch.bfloat16)
    bias = torch.zeros(N, device=device)
    res = torch.zeros(M, N, device=device, dtype=torch.bfloat16)
    for _ in range(3):
        ops.gemm(a, w, M, N, K, bias=bias, res=res)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        ops.gemm(a, w, M, N, K, bias=bias, res=res)
    e.record()
    torch.cuda.synchronize()
    t = s.elapsed_time(e) / iters * 1e-3
    ach = 2.0 * M * N * K / t / 1e12
    return {"kernel": "gemm_big_k<bf16,7,5>", "bound": "mfma", "achieved": round(ach, 1), "peak": MFMA_BF16_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": None, "launch_us": round(t * 1e6, 2),
            "shape": {"M": M, "N": N, "K": K, "dtype": "bfloat16", "epilogue": "bias+residual"}}


def _event_time(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def k2_roofline(device, C=1600, Ln=31):
    """K2: the pointwise-conv MLP of a stage-3 ConvNeXt block (convnext.py:41-45: Linear C->4C, GELU, Linear 4C->C with
    LayerScale + residual in the epilogues) on libmmhip's MFMA GEMM (these shapes take the large-tile kernel gemm_big_k, which
    splits K for the skinny 4C -> C product), bf16: SURVEY 8d flops 16*B*L*C^2 per block forward,
    at the config batch (M = B*L = 992 rows: too few tiles to fill 256 CUs) and at a saturating batch."""
    from multimodal_amd import ops
    out = {"kernel": "gemm_big_k<bf16> x2 (pwconv1+GELU, pwconv2+LayerScale+residual)", "bound": "mfma", "unit": "TFLOP/s",
           "peak": MFMA_BF16_PEAK_TFLOPS, "traffic": None, "shape": {"C": C, "L": Ln, "dtype": "bfloat16"}}
    w1 = (torch.randn(4 * C, C, device=device) * 0.02).to(torch.bfloat16)
    w2 = (torch.randn(C, 4 * C, device=device) * 0.02).to(torch.bfloat16)
    b1 = torch.zeros(4 * C, device=device)
    b2 = torch.zeros(C, device=device)
    gamma = torch.ones(C, device=device)
    for B in (32, 256):
        M = B * Ln
        xn = torch.randn(M, C, device=device).to(torch.bfloat16)
        res = torch.randn(M, C, device=device).to(torch.bfloat16)

        def mlp():
            a = ops.gemm(xn, w1, M, 4 * C, C, bias=b1, act=ops.ACT_GELU)
            return ops.gemm(a, w2, M, C, 4 * C, bias=b2, gamma=gamma, res=res)
        t = _event_time(mlp)
        ach = 16.0 * M * C * C / t / 1e12
        out[f"batch{B}"] = {"rows": M, "us_per_block_fwd": round(t * 1e6, 1), "achieved": round(ach, 1),
                            "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4)}
    out["achieved"], out["frac"] = out["batch256"]["achieved"], out["batch256"]["frac"]
    return out


def attn_roofline(device, B=32, H=16, d=100, N=190, T=94):
    """K3 / K3b at the local_config shape (S = 31+N+T = 315, prompt 221, stablemax, bf16, q/k pre-rotated as in the
    model): SURVEY 8d flops 4*B*S^2*E forward, 2.5x that backward (mask sparsity not discounted), against the dense bf16
    MFMA peak.  Per score the stablemax weight costs 5-6 vector instructions against 2 MFMAs' worth of matrix work per 64 scores,
    so vector issue, not the matrix cores, bounds these kernels (counters: profiles/r03_attn_d100_pmc.json)."""
    from multimodal_amd import ops
    E, S, P = H * d, 31 + N + T, 31 + N
    rot = int(d * 0.5)
    qkv = (torch.rand(B, S, 3, E, device=device) * 2 - 1).to(torch.bfloat16)
    inv = 1.0 / (10000 ** (torch.arange(0, rot, 2, device=device).float() / rot))
    cos_t, sin_t = ops.rope_tables(inv, S)
    q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    ops.rope_inplace(qkv.view(B, S, 3 * E)[:, :, :2 * E], 2 * H, rot, cos_t, sin_t)
    o, lse = ops.prefix_attn_fwd(q, k, v, H, 0, P, cos_t, sin_t, True, save_lse=True)
    do = torch.randn_like(o)
    tf = _event_time(lambda: ops.prefix_attn_fwd(q, k, v, H, 0, P, cos_t, sin_t, True, save_lse=True))
    tb = _event_time(lambda: ops.prefix_attn_bwd(q, k, v, o, do, lse, H, 0, P, cos_t, sin_t, True))
    fl = 4.0 * B * S * S * E
    af, ab = fl / tf / 1e12, 2.5 * fl / tb / 1e12
    counters = None
    if (B, H, d) == (32, 16, 100) and os.path.exists(os.path.join(ROOT, ATTN_PMC)):           # rocprofv3 --pmc passes over benchmarks/attn_pmc.py, same shape
        with open(os.path.join(ROOT, ATTN_PMC)) as fh:
            pm = json.load(fh)
        counters = {"source": ATTN_PMC + " (not re-measured in this run)"}
        for kname, e in pm.items():
            counters[kname] = {k: v for k, v in e.items() if k != "counters_per_launch" and k != "kernel_name"}
    return {"counters": counters, "kernel": ATTN_KERNELS_WIDE if d > 192 else ATTN_KERNELS, "bound": "mfma", "unit": "TFLOP/s",
            "peak": MFMA_BF16_PEAK_TFLOPS, "achieved": round(af, 1), "frac": round(af / MFMA_BF16_PEAK_TFLOPS, 4),
            "launch_us": round(tf * 1e6, 1), "bwd_achieved": round(ab, 1), "bwd_frac": round(ab / MFMA_BF16_PEAK_TFLOPS, 4),
            "bwd_launch_us": round(tb * 1e6, 1), "traffic": None,
            "algorithmic_bytes_fwd": 4 * B * S * E * 2, "reference_intermediate_bytes_fwd": 7 * B * H * S * S * 4,
            "shape": {"B": B, "H": H, "d": d, "S": S, "prompt": P, "dtype": "bfloat16", "stablemax": True}}


def decode_roofline(device, B=1024, H=8, d=128, Smax=238):
    """K4 at the nocat greedy shape, last position: algorithmic bytes = the K and V cache rows read once,
    B*H*2*(pos+1)*d*2, against HBM."""
    from multimodal_amd import ops
    E = H * d
    rot = int(d * 0.5)
    pos = Smax - 1
    kc = (torch.rand(B, H, Smax, d, device=device) - 0.5).to(torch.bfloat16)
    vc = (torch.rand(B, H, Smax, d, device=device) - 0.5).to(torch.bfloat16)
    qkv = (torch.rand(B, 3 * E, device=device) - 0.5).to(torch.bfloat16)
    inv = 1.0 / (10000 ** (torch.arange(0, rot, 2, device=device).float() / rot))
    cos_t, sin_t = ops.rope_tables(inv, Smax)
    t = _event_time(lambda: ops.decode_attn(qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:], kc, vc, H, rot, pos, cos_t, sin_t, False),
                    iters=50)
    nbytes = B * H * 2 * (pos + 1) * d * 2
    ach = nbytes / t / 1e9
    traffic = None              # PMC passes over benchmarks/decode_pmc.py (same shape), summarised by benchmarks/pmc_summary.py
    pmc = os.path.join(ROOT, K4_PMC)
    if (B, H, d, Smax) == (1024, 8, 128, 238) and os.path.exists(pmc):
        with open(pmc) as fh:
            traffic = int(json.load(fh)["traffic_bytes_per_launch"])
    return {"kernel": "decode_attn1_k", "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": traffic, "launch_us": round(t * 1e6, 2),
            "traffic_source": (K4_PMC + " (rocprofv3 --pmc passes over benchmarks/decode_pmc.py, same shape; not "
                               "re-measured in this run)") if traffic is not None else None,
            "algorithmic_bytes_per_launch": nbytes,
            "shape": {"B": B, "H": H, "d": d, "pos": pos, "dtype": "bfloat16", "softmax": True}}


def greedy_bench(device, world, rank):
    import multimodal_amd as mm
    c = CONFIGS["local_config_nocat"]
    B = max(1024 // world, 1)
    m = build_model(c, device).eval()
    ir, nmr, _ = synth_batch(c, B, device, 4321 + rank)
    steps = c["max_seq"]
    res = {}
    for name, dtype in (("bf16", torch.bfloat16), ("f32", torch.float32)):
        with mm.compute_dtype(dtype):
            m.generate(nmr, ir, max_len=steps, early_stop=False)      # warm-up at the measured shape (allocator, GEMM selection)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            toks = m.generate(nmr, ir, max_len=steps, early_stop=False)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            t1 = time.perf_counter()
            m.generate(nmr, ir, max_len=1, early_stop=False)          # prefill + first token only
            torch.cuda.synchronize()
            t_pre = time.perf_counter() - t1
        res[name] = (round(B * (toks.shape[1] - 1) / dt, 1), dt, t_pre)
    E, H, L = c["E"], c["H"], c["L"]
    P = 31 + c["max_nmr"]
    n_dec = steps - 1
    es = 2
    # decode-loop roofline (bf16): per step every layer reads the K and V cache rows 0..pos of every sample once and the
    # 12 E^2 weights once; achieved = those bytes over the measured time per decode step (whole loop minus the prefill)
    kv = sum(B * 2 * (P + 1 + t) * E * es for t in range(1, steps)) * L
    wts = 12 * E * E * es * L * n_dec
    t_loop = max(res["bf16"][1] - res["bf16"][2], 1e-9)
    ach = (kv + wts) / t_loop / 1e9
    return {"tokens_per_s_per_gpu": res["bf16"][0], "batch_per_gpu": B, "steps": steps, "config": "local_config_nocat", "dtype": "bf16",
            "kv_cache": True, "seconds": round(res["bf16"][1], 4), "prefill_seconds": round(res["bf16"][2], 4),
            "fp32": {"tokens_per_s_per_gpu": res["f32"][0], "seconds": round(res["f32"][1], 4),
                     "note": "the path on which token ids equal the reference's (tests/test_config_gpu.py)"},
            "decode_loop_roofline": {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                     "frac": round(ach / HBM_PEAK_GBPS, 4), "ms_per_decode_step": round(t_loop / n_dec * 1e3, 4),
                                     "kv_bytes_per_step_avg": kv // n_dec, "weight_bytes_per_step": wts // n_dec,
                                     "note": "K4 (the KV read, one pass, 0.76 of HBM alone) is about 0.56 ms of the 0.88 ms step; the rest is "
                                             "16 short latency-bound GEMMs per half batch and LayerNorm; the two half-batch loops are "
                                             "issued from prepared launches on two streams: GPU-bound"}}


def train_leg(c, device, batch, dtype, steps, warmup, rank=0, backend="auto"):
    """Extra single-rank measurement of the same train step at another per-GPU batch / compute dtype (SURVEY 8d:
    report the config batch and a saturating batch, bf16 and fp32)."""
    import multimodal_amd as mm
    from multimodal_amd.dp import DataParallel
    from multimodal_amd.optim import OrthoGradAdamW
    prev_backend = mm.get_gemm_backend()
    mm.set_gemm_backend(backend)           # "own": every dense product on libmmhip (no library GEMM on the path)
    with mm.compute_dtype(dtype):
        model = build_model(c, device).train()
        dp = DataParallel(model)
        opt = OrthoGradAdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, ortho=True,
                             ortho_eps=1e-8, rescale=True)
        if os.environ.get("MM_OPT_DEFER", "1") == "1":
            opt.defer_into_next_forward(model, model.decoder.layers)
        ir, nmr, tgt = synth_batch(c, batch, device, 99 + rank)
        crit = torch.nn.CrossEntropyLoss(ignore_index=0)

        def step():
            dp.zero_grad()
            logits = model(nmr, ir, tgt[:, :-1], None)
            crit(logits.reshape(-1, logits.size(-1)), tgt[:, 1:].reshape(-1)).backward()
            dp.finish()
            opt.step()
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        opt.flush_deferred(wait=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    del model, dp, opt
    torch.cuda.empty_cache()
    mm.set_gemm_backend(prev_backend)
    return {"spectra_per_s_per_gpu": round(batch / dt, 1), "ms_per_step": round(dt * 1e3, 2), "per_gpu_batch": batch,
            "dtype": str(dtype).replace("torch.", ""), "steps": steps, "gemm_backend": backend}


def graph_leg(device, name="local_config_nocat", batch=32, steps=20):
    """The same train step issued eagerly and replayed from ONE captured HIP graph (multimodal_amd.graph), on a config whose
    step is launch-bound at the config batch (the headline config is GPU-bound: its replay gains nothing)."""
    import multimodal_amd as mm
    from multimodal_amd.dp import DataParallel
    from multimodal_amd.graph import CapturedTrainStep
    from multimodal_amd.optim import OrthoGradAdamW
    c = CONFIGS[name]
    with mm.compute_dtype(torch.bfloat16):
        model = build_model(c, device).train()
        dp = DataParallel(model)
        opt = OrthoGradAdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, ortho=True,
                             ortho_eps=1e-8, rescale=True, capturable=True)
        ir, nmr, tgt = synth_batch(c, batch, device, 4242)
        crit = torch.nn.CrossEntropyLoss(ignore_index=0)

        def step():
            dp.zero_grad()
            logits = model(nmr, ir, tgt[:, :-1], None)
            loss = crit(logits.reshape(-1, logits.size(-1)), tgt[:, 1:].reshape(-1))
            loss.backward()
            dp.finish()
            opt.step()
            return loss.detach()

        def clock(fn):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / steps * 1e3
        for _ in range(3):
            step()
        eager = clock(step)
        cap = CapturedTrainStep(step, opt, warmup=1)
        replay = clock(cap)
    del model, dp, opt, cap
    torch.cuda.empty_cache()
    return {"config": name, "per_gpu_batch": batch, "dtype": "bfloat16", "eager_ms_per_step": round(eager, 3),
            "graph_replay_ms_per_step": round(replay, 3), "spectra_per_s_replay": round(batch / replay * 1e3, 1), "steps": steps}


def greedy_nocache(device, B=128):
    """The reference's decode loop (train_autoregressive.py:97-101: the whole decoder is re-run on the growing sequence
    every step) on the same device and kernels, beside the KV-cached generate, same batch: the speed-up of row K4."""
    import multimodal_amd as mm
    c = CONFIGS["local_config_nocat"]
    m = build_model(c, device).eval()
    ir, nmr, _ = synth_batch(c, B, device, 777)
    steps = c["max_seq"]
    with mm.compute_dtype(torch.bfloat16), torch.no_grad():
        m.generate(nmr, ir, max_len=steps, early_stop=False)      # warm-up at the measured shape
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.generate(nmr, ir, max_len=steps, early_stop=False)
        torch.cuda.synchronize()
        t_cache = time.perf_counter() - t0
        mem = m.encoder(None, ir, None)
        cur = torch.full((B, 1), 12, dtype=torch.long, device=device)
        for _ in range(3):                                         # warm-up of the recompute loop's first shapes
            m.decoder(cur, mem, nmr)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            logits = m.decoder(cur, mem, nmr)
            cur = torch.cat([cur, logits[:, -1].argmax(-1, keepdim=True)], 1)
        torch.cuda.synchronize()
        t_nocache = time.perf_counter() - t0
    return {"batch": B, "steps": steps, "kv_cache_tokens_per_s": round(B * steps / t_cache, 1),
            "recompute_tokens_per_s": round(B * steps / t_nocache, 1), "speedup": round(t_nocache / t_cache, 2)}


def test_config_leg(device, B=8):
    """BASELINE.json configs[0] (configs/test_config.yaml, E768 H4 L8 stablemax, batch 8): one forward over the full
    sequence and one KV-cached greedy decode of max_seq steps, fp32 (exact-f32 kernels: the path on which ids equal the
    reference's).  The reference runs this case on the CPU; here it is the same plumbing on the HIP path."""
    import multimodal_amd as mm
    c = CONFIGS["test_config"]
    m = build_model(c, device).eval()
    ir, nmr, tgt = synth_batch(c, B, device, 2024)
    steps = c["max_seq"]
    with mm.compute_dtype(torch.float32), torch.no_grad():
        tf = _event_time(lambda: m(nmr, ir, tgt[:, :-1], None), iters=5, warmup=2)
        m.generate(nmr, ir, max_len=steps, early_stop=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        toks = m.generate(nmr, ir, max_len=steps, early_stop=False)
        torch.cuda.synchronize()
        tg = time.perf_counter() - t0
    return {"config": "configs/test_config.yaml (E768 H4 d192 L8 stablemax)", "batch": B, "dtype": "float32",
            "forward_ms": round(tf * 1e3, 3), "forward_spectra_per_s": round(B / tf, 1),
            "greedy_steps": int(toks.shape[1] - 1), "greedy_seconds": round(tg, 4),
            "greedy_tokens_per_s": round(B * (toks.shape[1] - 1) / tg, 1)}


def cpu_baseline(c, sample_b=32, timed=3, warm=1):
    """The CPU oracle (oracle/restate.py, pinned against the reference) timed on the host cores: the train step
    (forward + CE + backward + OrthoGrad + AdamW) of the same shape at the config batch, one warm-up + `timed` steps."""
    from oracle import restate
    from oracle.shapes import state_shapes
    cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16") or 16))
    torch.set_num_threads(cores)
    cfg = restate.Cfg(embed_dim=c["E"], num_heads=c["H"], num_layers=c["L"], max_seq_length=c["max_seq"],
                      max_nmr_length=c["max_nmr"], max_memory_length=c["max_mem"], use_stablemax=c["sm"])
    g = torch.Generator().manual_seed(0)
    sd = {}
    for k, shp in state_shapes(cfg).items():
        if k.endswith("inv_freq"):
            continue
        if k.endswith("norm.weight") or ".1.weight" in k or ".0.2.weight" in k:
            t = torch.ones(shp)
        elif k.endswith("gamma"):
            t = torch.full(shp, 1e-6)
        elif k.endswith("bias"):
            t = torch.zeros(shp)
        else:
            t = torch.randn(shp, generator=g) * 0.02
        sd[k] = t.requires_grad_(True)
    params = list(sd.values())
    opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.01)
    ir, nmr, tgt = synth_batch(c, sample_b, "cpu", 1234)

    def step():
        loss = restate.train_loss(sd, cfg, nmr, ir, tgt, training=True)
        opt.zero_grad()
        loss.backward()
        restate.orthograd_project_(params, eps=1e-8)
        opt.step()
    for _ in range(warm):                             # warm-up (allocator, thread pool, oneDNN primitive caches)
        step()
    times = []
    for _ in range(timed):
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    dt = sorted(times)[len(times) // 2]
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            model = next((ln.split(":", 1)[1].strip() for ln in fh if ln.startswith("model name")), "unknown")
    except OSError:
        pass
    return {"value": round(sample_b / dt, 3), "unit": "spectra/s", "cores": cores, "kind": "port", "cpu_model": model,
            "host_cpus_visible": os.cpu_count(),
            "cores_note": "threads = the box's CPU share per GPU (OMP_NUM_THREADS as set by the pool), not the whole host",
            "step_seconds": [round(t, 2) for t in times],
            "sample": f"{warm} warm-up + {timed} timed train steps (fwd+CE+bwd+OrthoGrad+AdamW), median, fp32, local_config "
                      f"shape, batch {sample_b}, N=190 T=94, {dt:.1f} s per step"}


def spawn_ranks(a, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher above us: this process (which has not touched the GPU) starts
    one rank per GPU through torch.distributed.run as a CHILD process and passes its exit code on -- exactly the command
    the driver would have issued.  Never an exec: a process that has initialised HIP must not be replaced."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    log(f"spawning {a.gpus} ranks: {' '.join(cmd)}")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def dry_run(a, world, rank, json_out=sys.stdout):
    """--dry-run: the launch / rendezvous / exchange plumbing without a GPU (backend gloo, CPU tensors): each rank runs the
    production DataParallel wrapper (buckets, hook-driven all-reduce, finish) around a small stand-in module, and rank 0
    prints a JSON line of the same form.  Covers `--gpus N` in the CPU test-suite; measures nothing."""
    import torch.distributed as dist
    from multimodal_amd.dp import DataParallel
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(16, 64), torch.nn.GELU(), torch.nn.Linear(64, 8))
    dp = DataParallel(m, bucket_bytes=1 << 10, force_collectives=True) if world >= 1 and dist.is_initialized() else None
    opt = torch.optim.SGD(m.parameters(), lr=0.1)
    g = torch.Generator().manual_seed(100 + rank)
    x, y = torch.randn(8, 16, generator=g), torch.randint(0, 8, (8,), generator=g)
    if dist.is_initialized():
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        dp.zero_grad() if dp else opt.zero_grad()
        loss = torch.nn.functional.cross_entropy(m(x), y)
        (dp.scale_loss(loss) if dp else loss).backward()
        if dp:
            dp.finish()
        opt.step()
    if dist.is_initialized():
        dist.barrier()
    elapsed = time.perf_counter() - t0
    per_rank = [elapsed]
    if dist.is_initialized():
        tl = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(tl, torch.tensor([elapsed], dtype=torch.float64))
        per_rank = [float(t) for t in tl]
        chk = torch.cat([p.detach().reshape(-1) for p in m.parameters()])
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        assert torch.equal(lo, hi), "replicas diverged"
    if rank == 0:
        print(json.dumps({"metric": "train spectra/sec", "value": None, "unit": "spectra/s", "n_gpus": world, "steps": a.steps,
                          "warmup": a.warmup, "dry_run": True, "backend": a.backend, "rccl_ranks": world,
                          "allreduce_calls": dp.n_allreduce if dp else 0, "replicas_equal": True,
                          "per_rank_ms_per_step": [round(t / max(1, a.steps) * 1e3, 3) for t in per_rank]}), file=json_out, flush=True)


def rccl_world1_leg(c, device, steps=5):
    """N = 1 only: the SAME train step with the process group initialised on the nccl (= RCCL) backend and every bucket
    all-reduce forced through it (one rank; a second rank cannot share the device): proof, in the driver's own N = 1 run,
    that the communicator, the hook-driven asynchronous all-reduce on RCCL's stream and finish() execute on this GPU."""
    import socket
    import torch.distributed as dist
    import multimodal_amd as mm
    from multimodal_amd.dp import DataParallel
    from multimodal_amd.optim import OrthoGradAdamW
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    try:
        with mm.compute_dtype(torch.bfloat16):
            model = build_model(c, device).train()
            dp = DataParallel(model, force_collectives=True)
            opt = OrthoGradAdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, ortho=True,
                                 ortho_eps=1e-8, rescale=True)
            if os.environ.get("MM_OPT_DEFER", "1") == "1":
                opt.defer_into_next_forward(model, model.decoder.layers)
            ir, nmr, tgt = synth_batch(c, 32, device, 4)
            crit = torch.nn.CrossEntropyLoss(ignore_index=0)

            def step():
                dp.zero_grad()
                logits = model(nmr, ir, tgt[:, :-1], None)
                dp.scale_loss(crit(logits.reshape(-1, logits.size(-1)), tgt[:, 1:].reshape(-1))).backward()
                dp.finish()
                opt.step()
            for _ in range(3):
                step()
            dist.barrier()
            torch.cuda.synchronize()
            n0 = dp.n_allreduce
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            opt.flush_deferred(wait=True)
            dist.barrier()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            out = {"backend": dist.get_backend(), "ranks": dist.get_world_size(), "ms_per_step": round(dt * 1e3, 2),
                   "allreduce_calls_per_step": (dp.n_allreduce - n0) // steps, "buckets": len(dp.buckets.buckets),
                   "bucket_mib": [round(b["flat"].numel() * 4 / 2 ** 20, 1) for b in dp.buckets.buckets],
                   "overlap_hooks": bool(dp.overlap)}
        del model, dp, opt
        torch.cuda.empty_cache()
    finally:
        dist.destroy_process_group()
    return out


def run_leg(extras, key, fn, *args, **kw):
    """One extra leg: a Python-level failure in it is recorded under its key (and on stderr) instead of costing the whole
    line; the headline value, measured before any extra leg, is printed whatever happens here."""
    import traceback
    try:
        extras[key] = fn(*args, **kw)
    except Exception as ex:                                   # noqa: BLE001 -- the line must survive any leg
        traceback.print_exc(file=sys.stderr)
        extras[key] = {"error": f"{type(ex).__name__}: {ex}"}
        log(f"leg {key} FAILED: {extras[key]['error']}")
    try:
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    except Exception:                                         # noqa: BLE001
        pass
    return extras[key]


def run_extras(a, c, device, world, rank):
    """Roofline, greedy, other-config and CPU-baseline legs of the JSON line (each one guarded by run_leg)."""
    import torch.distributed as dist
    q = a.quick
    extras = {}
    roof = run_leg(extras, "roofline", k1_roofline, device)
    run_leg(extras, "roofline_bf16", k1_roofline, device, torch.bfloat16)
    lat = run_leg({}, "lat", k1_roofline, device, torch.float32, B=32, iters=50)
    if "error" not in roof and "error" not in lat:
        roof["config_batch_latency_us"] = lat["launch_us"]
    log("K1 roofline done; greedy leg")
    greedy = run_leg(extras, "greedy", greedy_bench, device, world, rank)
    if world > 1:
        # every rank enters the collective whether or not its own leg failed (a rank that skipped it would hang the others)
        ok = "error" not in greedy
        gt = torch.tensor([greedy["tokens_per_s_per_gpu"] if ok else 0.0, 1.0 if ok else 0.0], device=device, dtype=torch.float64)
        dist.all_reduce(gt, op=dist.ReduceOp.SUM)
        if ok:
            greedy["tokens_per_s"] = round(float(gt[0].item()), 1)
            greedy["ranks_reporting"] = int(gt[1].item())
    elif "error" not in greedy:
        greedy["tokens_per_s"] = greedy["tokens_per_s_per_gpu"]
    run_leg(extras, "roofline_mfma", gemm_roofline, device)
    run_leg(extras, "roofline_k2", k2_roofline, device)
    run_leg(extras, "roofline_attn", attn_roofline, device)
    run_leg(extras, "roofline_attn_d512", attn_roofline, device, B=32, H=4, d=512)
    run_leg(extras, "roofline_decode", decode_roofline, device)
    run_leg(extras, "roofline_k1_all_shapes", k1_all_shapes, device)
    log("kernel rooflines done")
    if world == 1:
        run_leg(extras, "greedy_vs_recompute", greedy_nocache, device)
        log("greedy recompute leg done; batch-256, fp32, own-back-end, real_config_mup and test_config legs")
        run_leg(extras, "train_batch256_bf16", train_leg, c, device, 256, torch.bfloat16, steps=2 if q else 4, warmup=1 if q else 2)
        run_leg(extras, "train_batch32_fp32", train_leg, c, device, 32, torch.float32, steps=2 if q else 5, warmup=1 if q else 2)
        own = run_leg(extras, "train_own_gemms_bf16", train_leg, c, device, a.batch, torch.bfloat16, steps=2 if q else 10,
                      warmup=1 if q else 3, backend="own")
        own["note"] = ("every dense product of the step on libmmhip (MM_GEMM_BACKEND=own): no hipBLASLt kernel on the path; "
                       "the headline uses the per-product faster one")
        # BASELINE.json configs[4]: the widened muP model (E2048 H4 d512 L8), same step, config batch
        mup = run_leg(extras, "train_real_config_mup_bf16", train_leg, CONFIGS["real_config_mup"], device, 32, torch.bfloat16,
                      steps=2 if q else 8, warmup=1 if q else 3)
        mup["config"] = "configs/real_config_mup.yaml (E2048 H4 d512 L8 stablemax, N=190 T=94 S=315)"
        # BASELINE.json configs[0]: test_config, batch 8, forward + greedy decode (fp32: the path whose ids equal the reference's)
        run_leg(extras, "test_config_forward_greedy", test_config_leg, device)
        log("captured-step leg")
        run_leg(extras, "hip_graph_step", graph_leg, device, steps=3 if q else 20)
        log("RCCL one-rank leg")
        run_leg(extras, "rccl_world1", rccl_world1_leg, c, device, steps=2 if q else 5)
    if rank == 0 and world == 1:
        log("CPU baseline (oracle, train steps on the host cores)")
        run_leg(extras, "cpu_baseline", cpu_baseline, c, sample_b=a.cpu_batch, timed=a.cpu_steps, warm=0 if q else 1)
        log("CPU baseline done")
    return extras


def assemble_line(a, c, world, elapsed, final_loss, rccl, opt_updates, extras):
    """The ONE JSON line (rank 0).  Takes plain values only, so tests/test_bench_cpu.py can call it without a GPU."""
    gb = a.batch * world
    out = {
        "metric": "train spectra/sec", "value": round(gb * a.steps / elapsed, 2), "unit": "spectra/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": f"configs/{a.config}.yaml train step (fwd + CE + bwd + grad all-reduce + OrthoGrad-AdamW), "
                               f"E{c['E']} H{c['H']} L{c['L']} {'stablemax' if c['sm'] else 'softmax'}, N={c['max_nmr']} "
                               f"T={c['max_seq'] - 1} S={31 + c['max_nmr'] + c['max_seq'] - 1}",
                   "global_batch": gb, "per_gpu_batch": a.batch, "parallelism": f"dp{world}",
                   "master_weights": "f32", "random_init": True, "optimizer_updates": opt_updates},
        "mean_loss": round(final_loss, 4), "rccl_ranks": world if world > 1 else 0, "rccl": rccl,
    }
    out.update(extras)
    failed = sorted(k for k, v in extras.items() if isinstance(v, dict) and "error" in v)
    if failed:
        out["failed_legs"] = failed
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="local_config")
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (configs/*.yaml: training.batch_size = 32)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-extras", action="store_true", help="skip roofline / greedy / cpu baseline legs")
    ap.add_argument("--cpu-steps", type=int, default=3, help="timed train steps of the CPU baseline (about 13 s each at batch 32)")
    ap.add_argument("--cpu-batch", type=int, default=32, help="batch of the CPU baseline's sample (config batch = 32)")
    ap.add_argument("--quick", action="store_true",
                    help="every extra leg at its smallest step count (tests/test_bench_gpu.py: the whole default path in seconds)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend: nccl (= RCCL, the product path) or gloo (CPU plumbing check, --dry-run)")
    ap.add_argument("--dry-run", action="store_true", help="rendezvous + DataParallel plumbing on CPU tensors, no GPU, no timing")
    a = ap.parse_args()

    import torch.distributed as dist
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(a, sys.argv[1:]))
    # stdout carries exactly ONE line, the JSON: native libraries (RCCL's version banner) write to file descriptor 1 behind
    # Python's back, so fd 1 is pointed at stderr for the whole run and the line goes out through a saved copy of it
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    json_out = os.fdopen(json_fd, "w")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if a.dry_run:
        if a.backend != "gloo":
            raise SystemExit("bench.py --dry-run runs on CPU tensors: use --backend gloo")
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            assert dist.get_world_size() == a.gpus
        dry_run(a, world, rank, json_out)
        if world > 1:
            dist.destroy_process_group()
        return
    if a.backend != "nccl":
        raise SystemExit("bench.py: measured runs use --backend nccl (RCCL); gloo is for --dry-run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        assert dist.get_world_size() == a.gpus and dist.get_backend() == "nccl"

    log(f"torch {torch.__version__} up, device {torch.cuda.get_device_name(local)}, world {world}")
    import multimodal_amd as mm
    from multimodal_amd.dp import DataParallel
    from multimodal_amd.optim import OrthoGradAdamW
    c = CONFIGS[a.config]
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    mm.set_compute_dtype(dt)
    model = build_model(c, device).train()
    dp = DataParallel(model)
    assert dp.world == world and (world == 1 or (dp.collective and dp.overlap and not dp.host_staged))
    # optimizer of configs/local_config.yaml: ortho_adamw, lr 1e-4, betas (0.9,0.999), wd 0.01, ortho eps 1e-8, rescale
    opt = OrthoGradAdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, ortho=True,
                         ortho_eps=1e-8, rescale=True)
    if os.environ.get("MM_OPT_OVERLAP"):      # optimizer-in-backward: correct (tested) but measured 2 % slower on one GPU
        dp.attach_optimizer(opt)
    elif os.environ.get("MM_OPT_DEFER", "1") == "1":
        # the decoder layers' updates run at the top of the next forward, beside the encoder's small kernels (bit-identical
        # parameters: tests/test_deferred_optimizer_gpu.py); the last one is flushed before the clock stops
        opt.defer_into_next_forward(model, model.decoder.layers)
    ir, nmr, tgt = synth_batch(c, a.batch, device, 1234 + rank)
    crit = torch.nn.CrossEntropyLoss(ignore_index=0)
    loss_acc = torch.zeros((), device=device)

    def step(i):
        for gparam in opt.param_groups:                       # linear warm-up (train_autoregressive.py:195-226)
            gparam["lr"] = 1e-4 * min(1.0, (i + 1) / 100.0)
        dp.zero_grad()
        logits = model(nmr, ir, tgt[:, :-1], None)
        loss = crit(logits.reshape(-1, logits.size(-1)), tgt[:, 1:].reshape(-1))
        dp.scale_loss(loss).backward()            # /world here: the summed all-reduce is the average, no division pass
        dp.finish()
        opt.step()
        loss_acc.add_(loss.detach())

    log("model built; warm-up")
    for i in range(a.warmup):
        step(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    log(f"timing {a.steps} steps")
    n_ar0 = dp.n_allreduce
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(a.warmup + i)
    opt.flush_deferred(wait=True)                 # the last step's deferred update belongs to the timed region
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    per_rank = [elapsed]
    if world > 1:
        tl = [torch.zeros(1, device=device, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(tl, torch.tensor([elapsed], device=device, dtype=torch.float64))
        per_rank = [float(t.item()) for t in tl]
        elapsed = max(per_rank)                   # MAX over ranks
    final_loss = float(loss_acc.item()) / max(1, a.steps + a.warmup)
    log(f"train leg: {elapsed / a.steps * 1e3:.2f} ms/step")
    rccl = {"backend": "nccl" if world > 1 else None, "ranks": world,
            "allreduce_calls_per_step": (dp.n_allreduce - n_ar0) // max(1, a.steps),
            "buckets": len(dp.buckets.buckets), "bucket_mib": [round(b["flat"].numel() * 4 / 2 ** 20, 1) for b in dp.buckets.buckets],
            "per_rank_ms_per_step": [round(t / a.steps * 1e3, 3) for t in per_rank]}

    # everything the JSON line says about the optimizer is read BEFORE the objects go away (round 2 lost its whole line to a
    # read of `opt` after `del opt`)
    opt_updates = ("decoder layers deferred into the next forward (bit-identical, tests/test_deferred_optimizer_gpu.py); "
                   "the last one flushed inside the timed region") if getattr(opt, "_defer_ids", None) else "all inside opt.step()"
    extras = {}
    if not a.no_extras:
        del opt, dp, step
        model = None
        torch.cuda.empty_cache()
        extras = run_extras(a, c, device, world, rank)

    if rank == 0:
        out = assemble_line(a, c, world, elapsed, final_loss, rccl, opt_updates, extras)
        print(json.dumps(out), file=json_out, flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
