"""Time the fused BertFusion kernels against the general 4-launch / 9-launch sequences at the bench shape (B=256, Lq=64, Nv=393)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mkg_analogy_amd import ops

B, Lq, Nv, H = 256, int(os.environ.get("LQ", 64)), 393, 768
Nvp = ((Nv + 63) // 64) * 64
dev = torch.device("cuda:0")
BF, F32 = torch.bfloat16, torch.float32
q = (0.5 * torch.randn(B * Lq, H, device=dev)).to(BF); v = (0.5 * torch.randn(B * Nv, H, device=dev)).to(BF)
dout = torch.randn(B * Lq, H, device=dev).to(BF)
out = torch.empty(B * Lq, H, device=dev, dtype=BF); probs = torch.zeros(B * Lq, Nvp, device=dev, dtype=BF)
dq = torch.empty_like(out); dv = torch.zeros(B * Nv, H, device=dev); dvb = torch.zeros(B * Nv, H, device=dev, dtype=BF)
scores = torch.empty(B * Lq, Nvp, device=dev); visT = torch.empty(B * H, Nvp, device=dev, dtype=BF)
dprobs = torch.empty(B * Lq, Nvp, device=dev); dsc = torch.empty(B * Lq, Nvp, device=dev, dtype=BF)
dscT, prT = torch.empty(B * Nv, Lq, device=dev, dtype=BF), torch.empty(B * Nv, Lq, device=dev, dtype=BF)
ctxT, dfT = torch.empty(B * H, Lq, device=dev, dtype=BF), torch.empty(B * H, Lq, device=dev, dtype=BF)
Mt = B * Lq


def fused_fwd(): ops.fusion_fwd(q, v, out, probs, B, Lq, Nv, H)
def fused_bwd(): ops.fusion_bwd(q, v, dout, probs, dq, dv, dvb, B, Lq, Nv, H)
def gen_fwd():
    ops.gemm_nt(q, v, scores, M=Lq, N=Nv, batch=B, stride_a=Lq * H, stride_b=Nv * H, stride_c=Lq * Nvp)
    ops.softmax_fwd(scores, probs, Mt, Nv)
    ops.transpose_bf16(v, visT, Nv, H, Nvp, batch=B, stride_i=Nv * H, stride_o=H * Nvp)
    ops.gemm_nt(probs, visT, out, M=Lq, N=H, batch=B, stride_a=Lq * Nvp, stride_b=H * Nvp, stride_c=Lq * H)
def gen_bwd():
    ops.gemm_nt(dout, v, dprobs, M=Lq, N=Nv, batch=B, stride_a=Lq * H, stride_b=Nv * H, stride_c=Lq * Nvp)
    ops.softmax_bwd(probs, dprobs, dsc, Mt, Nv)
    ops.gemm_nt(dsc, visT, dq, M=Lq, N=H, batch=B, stride_a=Lq * Nvp, stride_b=H * Nvp, stride_c=Lq * H)
    ops.transpose_bf16(dsc, dscT, Lq, Nv, Lq, batch=B, stride_i=Lq * Nvp, stride_o=Nv * Lq)
    ops.transpose_bf16(probs, prT, Lq, Nv, Lq, batch=B, stride_i=Lq * Nvp, stride_o=Nv * Lq)
    ops.transpose_bf16(q, ctxT, Lq, H, Lq, batch=B, stride_i=Lq * H, stride_o=H * Lq)
    ops.transpose_bf16(dout, dfT, Lq, H, Lq, batch=B, stride_i=Lq * H, stride_o=H * Lq)
    ops.gemm_nt(dscT, ctxT, dv, A2=prT, B2=dfT, M=Nv, N=H, batch=B, stride_a=Nv * Lq, stride_b=H * Lq, stride_c=Nv * H, stride_aux=Nv * H, res_f32=dv, C2=dvb)


def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n)
    return sorted(ts)[2]


names = sys.argv[1:] or ["fused_fwd", "gen_fwd", "fused_bwd", "gen_bwd"]
for nme in names:
    if Lq % 64 and nme == "gen_bwd": continue
    print(f"{nme:10s} {timeit(globals()[nme]) * 1000:8.1f} us")
