"""Per-layer inference kernels of the diffusion head at the policy's shape (B=1, 616 tokens, D=120, 8 heads): the
thread-per-channel forms against the matrix-core forms.  Run on the GPU box: `python tools/time_layer_kernels.py`."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvblox_mindmap_amd.diffuser_actor import fused_ops as FO  # noqa: E402
from nvblox_mindmap_amd.diffuser_actor import layers as Ly  # noqa: E402


def timed(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()  # replay from a graph: launch overhead out of the picture, as in the sampler
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
    torch.cuda.synchronize()
    g.replay()
    a.record()
    g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


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
    with torch.no_grad():
        q, k, v = FO.qkv_block(x, ss, A.q_proj, A.kv_proj, rot)
        qh, kh, vt = FO.qkv_heads(x, ss, A.q_proj, A.kv_proj, rot, H)
        att = FO.attention_small(q, k, v, None, H)
        print(f"qkv_block        {timed(lambda: FO.qkv_block(x, ss, A.q_proj, A.kv_proj, rot)):7.2f} us")
        print(f"qkv_heads (mfma) {timed(lambda: FO.qkv_heads(x, ss, A.q_proj, A.kv_proj, rot, H)):7.2f} us")
        print(f"attention_small  {timed(lambda: FO.attention_small(q, k, v, None, H)):7.2f} us")
        print(f"attention_heads  {timed(lambda: FO.attention_heads(qh, kh, vt, None, L, L)):7.2f} us")
        print(f"out_ffn_block    {timed(lambda: FO.out_ffn_block(att, x, A.out_proj, blk.norm, ss, ffw.fc1, ffw.fc2, ffw.norm)):7.2f} us")
        print(f"out_ffn_mfma     {timed(lambda: FO.out_ffn_mfma(att, x, A.out_proj, blk.norm, ss, ffw.fc1, ffw.fc2, ffw.norm)):7.2f} us")

        def layer_old():
            q, k, v = FO.qkv_block(x, ss, A.q_proj, A.kv_proj, rot)
            a = FO.attention_small(q, k, v, None, H)
            return FO.out_ffn_block(a, x, A.out_proj, blk.norm, ss, ffw.fc1, ffw.fc2, ffw.norm)

        def layer_new():
            qh, kh, vt = FO.qkv_heads(x, ss, A.q_proj, A.kv_proj, rot, H)
            a = FO.attention_heads(qh, kh, vt, None, L, L)
            return FO.out_ffn_mfma(a, x, A.out_proj, blk.norm, ss, ffw.fc1, ffw.fc2, ffw.norm)

        print(f"layer, channel kernels {timed(layer_old):7.2f} us")
        print(f"layer, mfma kernels    {timed(layer_new):7.2f} us")


if __name__ == "__main__":
    main()
