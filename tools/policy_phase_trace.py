"""Where the time goes INSIDE the per-layer inference kernels of the diffusion head (k_qkv_heads, k_attention_heads,
k_out_ffn_mfma) at the policy's shape: phase marks of thread 0 of every workgroup (100 MHz wall clock), relative to the
launch's first workgroup start.  Needs the instrumented build:
    make -C nvblox_mindmap_amd/csrc WG_TRACE=1 OUT=../libmmfusion_trace.so BUILD=_build_trace
    MMF_LIB=libmmfusion_trace.so python tools/policy_phase_trace.py
The layer's three kernels run back to back (as in the sampler), the buffer is read after each repetition."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvblox_mindmap_amd import _lib  # noqa: E402
from nvblox_mindmap_amd.diffuser_actor import fused_ops as FO  # noqa: E402
from nvblox_mindmap_amd.diffuser_actor import layers as Ly  # noqa: E402

BASES = {"k_qkv_heads": (0, 5, ["entry", "x tile in LDS", None, "GEMM halves in LDS", "end"]),
         "k_attention_heads": (256, 6, ["entry", "K/V/q in", "S done", "PV done", "merged (barrier)", "end"]),
         "k_out_ffn_mfma": (1024, 7, ["entry", "att tile + operands in LDS", "GEMM1 + LN1", "GEMM2 + u tile", "GEMM3 + LN2", "(fused) projections", "(fused) stores"])}


def main():
    torch.manual_seed(0)
    B, L, D, H = 1, int(os.environ.get("L", "616")), 120, 8
    blk = Ly.AttentionBlock(D, H, 0.0, use_adaln=True).cuda().eval()
    ffw = Ly.FeedForwardBlock(D, D, 0.0, use_adaln=True).cuda().eval()
    A = blk.attn
    x = torch.randn(B, L, D, device="cuda")
    ss = 0.3 * torch.randn(B, 2 * D, device="cuda")
    rot = Ly.rotary3d(torch.rand(B, L, 3, device="cuda"), D)
    rot = (rot[0].expand(B, L, D).contiguous(), rot[1].expand(B, L, D).contiguous())
    cap = 6 * 8192 + 8 * 2048
    buf = torch.zeros(3 * cap, dtype=torch.int64, device="cuda")
    _lib.check(_lib.lib().mmf_debug_wg_trace(_lib.dptr(buf), cap), "mmf_debug_wg_trace")
    acc = {k: [] for k in BASES}
    spans = {k: [] for k in BASES}
    try:
        with torch.no_grad():
            def layer(xin):
                qh, kh, vt = FO.qkv_heads(xin, ss, A.q_proj, A.kv_proj, rot, H)
                a = FO.attention_heads(qh, kh, vt, None, L, L)
                if os.environ.get("FUSED"):
                    return FO.out_ffn_qkv(a, xin, A.out_proj, blk.norm, ss, ffw.fc1, ffw.fc2, ffw.norm, ss, A.q_proj, A.kv_proj, rot, H)[0]
                return FO.out_ffn_mfma(a, xin, A.out_proj, blk.norm, ss, ffw.fc1, ffw.fc2, ffw.norm)

            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                y = layer(x)
                with torch.cuda.graph(g, stream=s):
                    y = x
                    for _ in range(3):  # three layers back to back; the marks of the LAST one survive
                        y = layer(y)
            torch.cuda.synchronize()
            for rep in range(30):
                buf.zero_()
                torch.cuda.synchronize()
                g.replay()
                torch.cuda.synchronize()
                raw = buf.cpu().numpy()[3 * 6 * 8192:].reshape(-1, 8)
                for name, (base, n, _) in BASES.items():
                    nxt = min([b for b, _, _ in BASES.values() if b > base] + [raw.shape[0]])
                    r = raw[base:nxt, :n]
                    r = r[r[:, 0] != 0]
                    if r.shape[0] == 0:
                        continue
                    t0 = r[:, 0].min()
                    acc[name].append(np.median((r - r[:, :1]) / 100.0, axis=0))  # per-workgroup phase times, median workgroup
                    spans[name].append(((r[:, 0].max() - t0) / 100.0, (r[:, n - 1].max() - t0) / 100.0, r.shape[0]))
    finally:
        _lib.lib().mmf_debug_wg_trace(None, 0)
    for name, (base, n, labels) in BASES.items():
        if not acc[name]:
            continue
        med = np.median(np.stack(acc[name][5:]), axis=0)
        sp = np.median(np.array(spans[name][5:]), axis=0)
        end = f"{sp[1]:.2f}" if 0 <= sp[1] < 1e6 else "?"
        print(f"{name}: {int(sp[2])} workgroups, last start +{sp[0]:.2f} us, last end +{end} us after the first start")
        prev = 0.0
        for i in range(n):
            if labels[i] is None or med[i] < 0 or med[i] > 1e6:  # a mark this kernel (or this variant of it) does not set
                continue
            print(f"    {labels[i]:<28} +{med[i]:6.2f} us" + (f"   (step {med[i] - prev:5.2f})" if i else ""))
            prev = med[i]


if __name__ == "__main__":
    main()
