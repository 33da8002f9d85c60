#!/usr/bin/env python3
"""Bitwise repeatability of the decoder layers' OrthoGrad-AdamW update when it runs on the side stream beside an encoder
forward on the main stream (what defer_into_next_forward does) against the same update run alone."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import multimodal_amd as mm  # noqa: E402
from multimodal_amd import functional as FN  # noqa: E402
from multimodal_amd.dp import DataParallel  # noqa: E402
from multimodal_amd.optim import OrthoGradAdamW  # noqa: E402
dev = torch.device("cuda:0")
c = bench.CONFIGS["local_config"]
mm.set_compute_dtype(torch.bfloat16)
model = bench.build_model(c, dev).train()
dp = DataParallel(model)
opt = OrthoGradAdamW(model.parameters(), lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, ortho=True, ortho_eps=1e-8, rescale=True)
ir, nmr, tgt = bench.synth_batch(c, 32, dev, 1234)
crit = torch.nn.CrossEntropyLoss(ignore_index=0)
for _ in range(2):                      # two real steps: moments are non-zero, gradients sit in their bucket slots
    dp.zero_grad()
    logits = model(nmr, ir, tgt[:, :-1], None)
    loss = crit(logits.reshape(-1, logits.size(-1)), tgt[:, 1:].reshape(-1))
    dp.scale_loss(loss).backward()
    dp.finish()
    opt.step()
dp.zero_grad()
logits = model(nmr, ir, tgt[:, :-1], None)
crit(logits.reshape(-1, logits.size(-1)), tgt[:, 1:].reshape(-1)).backward()
dp.finish()
torch.cuda.synchronize()
layers = list(model.decoder.layers)
ps = [p for l in layers for p in l.parameters()]
names = [f"L{i}.{n}" for i, l in enumerate(layers) for n, _ in l.named_parameters()]
snap = [(p.detach().clone(), opt.state[p]["exp_avg"].clone(), opt.state[p]["exp_avg_sq"].clone(), p.grad.clone()) for p in ps]   # the update rewrites g too (g_orth)


def restore():
    for p, (a, m, v, g) in zip(ps, snap):
        p.data.copy_(a); opt.state[p]["exp_avg"].copy_(m); opt.state[p]["exp_avg_sq"].copy_(v); p.grad.copy_(g)


if os.environ.get("OWN_WS") == "1":                 # every layer's launch sequence gets a workspace of its own
    from multimodal_amd import ops as _ops
    _orig_ws = _ops.workspace
    _cnt = [0]
    def _ws(nbytes, device, tag="ws"):
        if tag == "optim":
            _cnt[0] += 1
            return _orig_ws(nbytes, device, f"optim{_cnt[0] % 8}")
        return _orig_ws(nbytes, device, tag)
    _ops.workspace = _ws


def update(beside):
    restore()
    torch.cuda.synchronize()
    opt._step = 3
    opt.begin_step(); opt._step = 3
    main, side = FN.enter_side(dev)
    with torch.cuda.stream(side):
        for i, l in enumerate(layers):
            opt.step_subset(list(l.parameters()), ("chk", i))
    if beside == "encoder":
        with torch.no_grad():
            for _ in range(2):
                model.encoder(None, ir, None)
    elif beside.startswith("stage"):
        i = int(beside[5:])
        with torch.no_grad():
            for _ in range(6):
                y = STAGE_IN[i]
                for blk in bb.stages[i]:
                    y = blk.forward_blc(y)
    elif beside.startswith("down"):
        i = int(beside[4:])
        with torch.no_grad():
            for _ in range(20):
                ds = bb.downsample_layers[i]
                cw = ds[3].weight
                w2 = cw.permute(0, 2, 1).reshape(cw.shape[0], -1)
                FN.LNLinear.apply(STAGE_OUT[i - 1], ds[1].weight, ds[1].bias, w2, ds[3].bias, 2)
    elif beside == "stem":
        with torch.no_grad():
            for _ in range(20):
                stem = bb.downsample_layers[0]
                FN.StemLN.apply(ir.float().contiguous(), stem[0].weight, stem[0].bias, stem[2].weight, stem[2].bias, torch.bfloat16)
    elif beside == "head":
        with torch.no_grad():
            for _ in range(20):
                FN.LNLinear.apply(STAGE_OUT[3], bb.norm.weight, bb.norm.bias, bb.head.weight, bb.head.bias, 1)
    elif beside == "ln":
        for _ in range(60):
            _ops.layernorm_fwd(STAGE_OUT[1], bb.downsample_layers[2][1].weight, bb.downsample_layers[2][1].bias)
    elif beside == "gemmk":
        for _ in range(60):
            _ops.gemm(GA, GW, GA.shape[0], GW.shape[0], GA.shape[1], bias=GB)
    elif beside == "k1":
        blk = bb.stages[1][0]
        for _ in range(60):
            _ops.dwconv7_ln_fwd(STAGE_IN[1], blk.dwconv.weight, blk.dwconv.bias, blk.norm.weight, blk.norm.bias)
    elif beside == "gemmbig":
        for _ in range(30):
            _ops.gemm(BA, BW, BA.shape[0], BW.shape[0], BA.shape[1], bias=BB)
    elif beside == "attn":
        for _ in range(30):
            _ops.prefix_attn_fwd(AQ[:, :, 0], AQ[:, :, 1], AQ[:, :, 2], 16, 0, 221, ACOS, ASIN, True, save_lse=True)
    elif beside == "add":
        for _ in range(40):
            BIG.add_(1)
    elif beside == "mm":
        for _ in range(20):
            torch.mm(A_, B_)
    elif beside == "small":
        for _ in range(400):
            SMALL.add_(1)
    FN.join_side_stream()
    torch.cuda.synchronize()
    opt._begun = False
    global LAST_WS
    if os.environ.get("OWN_WS") == "1":
        with torch.cuda.stream(FN.side_stream(dev)):
            LAST_WS = [_orig_ws(1, dev, f"optim{k}").clone() for k in range(8)]
        torch.cuda.synchronize()
    global LAST_AUX
    LAST_AUX = [(opt.state[p]['exp_avg'].clone(), opt.state[p]['exp_avg_sq'].clone(), p.grad.clone()) for p in ps]
    return [p.detach().clone() for p in ps]


BIG = torch.zeros(64 << 20, dtype=torch.float32, device=dev)
SMALL = torch.zeros(4096, dtype=torch.float32, device=dev)
A_ = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
B_ = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
bb = model.encoder.ir_backbone
STAGE_IN, STAGE_OUT = {}, {}
with torch.no_grad():
    stem = bb.downsample_layers[0]
    y = FN.StemLN.apply(ir.float().contiguous(), stem[0].weight, stem[0].bias, stem[2].weight, stem[2].bias, torch.bfloat16)
    for i in range(4):
        if i > 0:
            ds = bb.downsample_layers[i]
            cw = ds[3].weight
            w2 = cw.permute(0, 2, 1).reshape(cw.shape[0], -1)
            y = FN.LNLinear.apply(y, ds[1].weight, ds[1].bias, w2, ds[3].bias, 2)
        STAGE_IN[i] = y
        for blk in bb.stages[i]:
            y = blk.forward_blc(y)
        STAGE_OUT[i] = y
torch.cuda.synchronize()
from multimodal_amd import ops as _ops
GA = torch.randn(3968, 256, device=dev, dtype=torch.bfloat16)
GW = torch.randn(1024, 256, device=dev, dtype=torch.bfloat16)
GB = torch.zeros(1024, device=dev)
BA = torch.randn(10080, 1600, device=dev, dtype=torch.bfloat16)
BW = torch.randn(1600, 1600, device=dev, dtype=torch.bfloat16)
BB = torch.zeros(1600, device=dev)
AQ = (torch.rand(32, 315, 3, 1600, device=dev) - 0.5).to(torch.bfloat16)
ACOS, ASIN = _ops.rope_tables(1.0 / (10000 ** (torch.arange(0, 50, 2, device=dev).float() / 50)), 315)
ref = update("")
again = update("")
print("alone vs alone: tensors differing:", len([n for n, a, b in zip(names, ref, again) if not torch.equal(a, b)]), flush=True)
REF_WS = LAST_WS if os.environ.get("OWN_WS") == "1" else None
REF_AUX = LAST_AUX
for rep in range(int(os.environ.get("REPS", "6"))):
    got = update(os.environ.get("BESIDE", "encoder"))
    diff = [(n, int((a != b).sum())) for n, a, b in zip(names, ref, got) if not torch.equal(a, b)]
    print("beside the encoder forward, run", rep, ": tensors differing:", len(diff), diff[:8], flush=True)
    if REF_WS is not None:
        for k in range(8):
            a, b = REF_WS[k].view(torch.int32), LAST_WS[k].view(torch.int32)
            d_ = (a != b).nonzero().flatten()
            if len(d_):
                c0 = int(d_[0]) // 3
                fa, fb = a.view(torch.float32), b.view(torch.float32)
                print("    layer", k, "chunk", c0, "ref (ww,wg,gg)", fa[3 * c0: 3 * c0 + 3].tolist(), "got", fb[3 * c0: 3 * c0 + 3].tolist(), flush=True)
                # which tensor / offset is chunk c0 of layer k?  recompute its sums from the snapshot in float64
                lps = list(layers[k].parameters()) if False else None
                print("    workspace", k, "differs at int32 offsets", d_[:10].tolist(), "count", len(d_), "values ref/got", a[d_[:4]].view(torch.float32).tolist(), b[d_[:4]].view(torch.float32).tolist(), flush=True)
    for n, a, b in zip(names, ref, got):
        if not torch.equal(a, b):
            idx = (a.flatten() != b.flatten()).nonzero().flatten()
            ch = torch.unique(idx // 65536, return_counts=True)
            rel = ((a.flatten()[idx] - b.flatten()[idx]).abs() / a.flatten()[idx].abs().clamp_min(1e-30)).max()
            k_ = names.index(n)
            p0, m0, v0, g0 = (t.flatten() for t in snap[k_])
            m1, v1, g1 = opt.state[ps[k_]]["exp_avg"].flatten(), opt.state[ps[k_]]["exp_avg_sq"].flatten(), ps[k_].grad.flatten()
            print("      same tensor: m differs in", int((REF_AUX[k_][0] != LAST_AUX[k_][0]).sum()), "v differs in", int((REF_AUX[k_][1] != LAST_AUX[k_][1]).sum()), "g_orth differs in", int((REF_AUX[k_][2] != LAST_AUX[k_][2]).sum()), flush=True)
            for j in idx[:2].tolist():
                print("      elem", j, "p0 %.9g g0 %.9g m0 %.9g v0 %.9g | ref p %.9g got p %.9g | got m %.9g v %.9g g %.9g" % (float(p0[j]), float(g0[j]), float(m0[j]), float(v0[j]), float(a.flatten()[j]), float(b.flatten()[j]), float(m1[j]), float(v1[j]), float(g1[j])), flush=True)
            print("   ", n, "numel", a.numel(), "chunks touched", ch[0].tolist()[:20], "counts", ch[1].tolist()[:20], "max rel diff %.3g" % float(rel), flush=True)
            break
