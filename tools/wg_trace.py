"""Per-workgroup timeline of the fused frame kernels (mmf_debug_wg_trace): for every role of every launch of one frame, the
first start, the last start, the last end and the mean / longest workgroup, relative to the frame's first record.
Needs the instrumented build of the library (the default build compiles the hooks out).  On the GPU box:
    make -C nvblox_mindmap_amd/csrc WG_TRACE=1 OUT=../libmmfusion_trace.so BUILD=_build_trace
    MMF_LIB=libmmfusion_trace.so python tools/wg_trace.py [frames]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402
from nvblox_mindmap_amd import _lib  # noqa: E402
from nvblox_mindmap_amd import synthetic as S  # noqa: E402
from nvblox_mindmap_amd.mapping.helpers.nvblox_mapping_helpers import get_nvblox_mapper  # noqa: E402
from nvblox_mindmap_amd.mapping.nvblox_mapper_constants import NvbloxMappingCfg  # noqa: E402

NAMES = {10: "k_front raycast", 11: "k_front mask rows", 12: "k_front decay", 20: "allocation workgroups", 21: "mask cols",
         30: "TSDF pass (existing blocks)", 31: "TSDF pass (new blocks)", 40: "k_sphere_alloc allocation", 41: "k_sphere_alloc trace", 50: "k_app_frame (deferred: role of k_front_app)", 60: "k_feature_flat (deferred: role of k_sphere_alloc_flat)"}


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    nframes = int(args[0]) if args else 3
    deferred = "--deferred" in sys.argv  # the software-pipelined stream: a call = 3 launches, the previous frame's tail inside
    dev = torch.device("cuda", 0)
    cfg = S.StreamConfig(hole_mode="patches")
    mcfg = NvbloxMappingCfg("DRILL_IN_BOX")
    frames = B.build_stream(cfg, 40, 64, dev)
    m = get_nvblox_mapper(mcfg, feature_channels=64)
    m.set_deferred_feature_rows(deferred)
    for i in range(24):
        B.step(m, mcfg, frames[i])
    torch.cuda.synchronize()
    phases = "--phases" in sys.argv  # in-workgroup phase marks of the TSDF pairs role (thread 0's block): slots 6 .. 8
    cap = (9 if phases else 6) * 8192
    buf = torch.zeros(3 * cap, dtype=torch.int64, device=dev)
    _lib.check(_lib.lib().mmf_debug_wg_trace(_lib.dptr(buf), cap), "mmf_debug_wg_trace")
    try:
        for i in range(24, 24 + nframes):
            buf.zero_()
            torch.cuda.synchronize()
            B.step(m, mcfg, frames[i])
            torch.cuda.synchronize()
            rec = buf.cpu().numpy().reshape(cap, 3)
            if phases:
                a, b, c = rec[6 * 8192:7 * 8192], rec[7 * 8192:8 * 8192], rec[8 * 8192:9 * 8192]
                rec = rec[:6 * 8192]
                for kind, name in ((2, "integrated this frame (cand)"), (1, "looked at only (appearance flag)")):
                    sel = a[:, 0] == kind
                    if not sel.any():
                        continue
                    t_begin, t_known, t_vox, t_loop, t_store, t_end = a[sel, 1], a[sel, 2], b[sel, 0], b[sel, 1], b[sel, 2], c[sel, 0]
                    us = lambda x: x / 100.0
                    tot = us(t_end - t_begin)
                    print(f"  TSDF pairs, blocks {name}: n={int(sel.sum())}  whole workgroup mean {tot.mean():.2f} us (p90 {np.percentile(tot, 90):.2f}, max {tot.max():.2f})")
                    for label, d in (("list -> slot -> key -> raycast flag known", us(t_known - t_begin)), ("voxels arrived (16-byte loads)", us(t_vox - t_known)),
                                     ("voxel loop (4 voxels: projection, depth taps, update)", us(t_loop - t_vox)), ("stores acknowledged", us(t_store - t_loop)),
                                     ("reductions + barrier + summary writes", us(t_end - t_store))):
                        print(f"      {label:56s} mean {d.mean():5.2f}  p90 {np.percentile(d, 90):5.2f}  max {d.max():5.2f} us")
            if "--starts" in sys.argv:  # when do the workgroups of launch 2 (k_alloc_tsdf) start, by dispatch index?
                blk = rec[2 * 8192:3 * 8192]  # role ids 30 / 31 share the slot range of ids 3x; 2x: the range before
                for lo, name in ((1, "ids 2x (allocation, mask columns)"), (2, "ids 3x (TSDF pairs, new-block waiters)")):
                    r = rec[lo * 8192:(lo + 1) * 8192]
                    idx = np.nonzero(r[:, 0])[0]
                    if len(idx) == 0:
                        continue
                    t00 = rec[rec[:, 0] != 0][:, 1].min()
                    st = (r[idx, 1] - t00) / 100.0
                    en = (r[idx, 2] - t00) / 100.0
                    print(f"  launch-2 {name}: {len(idx)} workgroups, blockIdx {idx.min()}..{idx.max()}")
                    for a in range(0, len(idx), max(len(idx) // 12, 1)):
                        b = min(a + max(len(idx) // 12, 1), len(idx))
                        print(f"    blockIdx {idx[a]:5d}..{idx[b - 1]:5d}: start {st[a:b].min():5.1f} .. {st[a:b].max():5.1f} (median {np.median(st[a:b]):5.1f})  end median {np.median(en[a:b]):5.1f} max {en[a:b].max():5.1f}")
                    late = idx[st > np.median(st) + 3.0]
                    if len(late):
                        print(f"    {len(late)} start > 3 us after the median; blockIdx mod 8 histogram {np.bincount(late % 8, minlength=8).tolist()}, first {late[:12].tolist()}")
            rec = rec[rec[:, 0] != 0]
            extra = rec[:, 0] >> 8
            rec[:, 0] &= 0xff
            ap = rec[:, 0] == 50
            if ap.any():  # k_app_frame: survivors / new-block flags of the workgroup's block ride in the record id
                ex, dur = extra[ap], (rec[ap, 2] - rec[ap, 1]) / 100.0
                surv, fnew, cnew, idle = ex & 0x3ff, (ex >> 10) & 1, (ex >> 11) & 1, ex == 0xfff
                order = np.argsort(-dur)[:8]
                print("  k_app_frame, slowest workgroups (us / survivors / feature block new / colour block new):",
                      ", ".join(f"{dur[o]:.1f}/{surv[o]}/{fnew[o]}/{cnew[o]}" for o in order))
                for name, sel in (("new feature blocks", (fnew == 1) & ~idle), ("old feature blocks", (fnew == 0) & ~idle), ("no block", idle)):
                    if sel.any():
                        print(f"    {name}: n={int(sel.sum())} mean {dur[sel].mean():.2f} us, mean survivors {surv[sel].mean():.0f}")
                live = ~idle
                if live.sum() > 10:
                    print(f"    correlation(duration, survivors) = {np.corrcoef(dur[live], surv[live])[0, 1]:.2f}")
            tr = rec[:, 0] == 41
            if tr.any():  # sphere trace: longest wide / narrow iteration counts per workgroup ride in the record id
                wide, narrow, dur = extra[tr] & 0xfff, (extra[tr] >> 12) & 0xfff, (rec[tr, 2] - rec[tr, 1]) / 100.0
                hi = extra[tr] >> 24  # narrow rounds of the patch's 16 rays by how they ended, and their steps
                miss, full, plain, fin, steps = hi & 63, (hi >> 6) & 63, (hi >> 12) & 63, (hi >> 18) & 63, (hi >> 24) & 255
                nr = miss + full + plain + fin
                if nr.sum() > 0:
                    print(f"  narrow rounds of all rays: {int(nr.sum())} in {int((nr > 0).sum())} patches; ended by: voxel not among the 16 probes {miss.sum() / nr.sum():.2f}, "
                          f"all 16 steps taken {full.sum() / nr.sum():.2f}, a plain step (back to wide) {plain.sum() / nr.sum():.2f}, ray finished {fin.sum() / nr.sum():.2f}; "
                          f"steps per round {steps.sum() / max(nr.sum(), 1):.1f} (step counts saturate at 255 per patch)")
                if len(dur) > 50:  # what a round costs: least squares of the patch duration on its longest ray's round counts
                    A = np.stack([np.ones_like(dur), wide.astype(np.float64), narrow.astype(np.float64)], axis=1)
                    coef, *_ = np.linalg.lstsq(A, dur, rcond=None)
                    res = dur - A @ coef
                    print(f"  sphere trace: duration ~ {coef[0]:.2f} + {coef[1]:.2f} x wide rounds + {coef[2]:.2f} x narrow rounds us (rms residual {np.sqrt((res ** 2).mean()):.2f}); "
                          f"histogram of (wide, narrow): " + ", ".join(f"({a},{b}): {int(((wide == a) & (narrow == b)).sum())} at {dur[(wide == a) & (narrow == b)].mean():.1f}"
                                                                      for a in range(0, 7) for b in range(0, 5) if ((wide == a) & (narrow == b)).sum() >= 10))
                order = np.argsort(-dur)[:6]
                print("  sphere trace, slowest workgroups (us, wide iterations, narrow iterations):",
                      ", ".join(f"{dur[o]:.1f}/{wide[o]}/{narrow[o]}" for o in order),
                      f"| mean iterations wide {wide.mean():.1f} narrow {narrow.mean():.1f}")
            t0 = rec[:, 1].min()
            print(f"frame {i}: {len(rec)} workgroup records, span {(rec[:, 2].max() - t0) / 100.0:.1f} us")
            if "--waiters" in sys.argv:  # every new-block waiter / allocation workgroup of launch 2: start -> end, relative to the launch's first start
                l2 = rec[(rec[:, 0] >= 20) & (rec[:, 0] < 40)]
                if len(l2):
                    t2 = l2[:, 1].min()
                    for rid in (20, 31):
                        r = l2[l2[:, 0] == rid]
                        print(f"    {NAMES[rid]}: " + " ".join(f"{(a - t2) / 100.0:.1f}-{(b - t2) / 100.0:.1f}" for a, b in sorted(zip(r[:, 1].tolist(), r[:, 2].tolist()), key=lambda x: x[1])))
                    pr = l2[l2[:, 0] == 30]
                    en = np.sort((pr[:, 2] - t2) / 100.0)
                    print(f"    existing-block pairs: ends at percentiles 50/90/99/100 = {np.percentile(en, 50):.1f} / {np.percentile(en, 90):.1f} / {np.percentile(en, 99):.1f} / {en.max():.1f} us; "
                          f"{int(((pr[:, 1] - t2) / 100.0 > 3.0).sum())} of {len(pr)} start later than 3 us")
                    pass
            for rid in sorted(set(rec[:, 0].tolist()), key=lambda r: rec[rec[:, 0] == r, 1].min()):
                r = rec[rec[:, 0] == rid]
                st, en = (r[:, 1] - t0) / 100.0, (r[:, 2] - t0) / 100.0
                d = en - st
                print(f"  {NAMES.get(rid, rid):28s} n={len(r):5d}  first start {st.min():6.1f}  last start {st.max():6.1f}  last end {en.max():6.1f}"
                      f"  mean wg {d.mean():5.2f}  longest wg {d.max():5.2f} us")
    finally:
        _lib.check(_lib.lib().mmf_debug_wg_trace(None, 0), "mmf_debug_wg_trace")


if __name__ == "__main__":
    main()
