#!/usr/bin/env python3
"""Race screen of every conv entry of the shipped launch-parameter table at its own shape: REPS launches each into NaN-filled outputs
(other kernels in between), every launch compared BIT FOR BIT with the first — the forward / data-gradient / inference kernels are
deterministic, so any difference is a synchronisation error (a missing wait, a hazard), whatever a reference would say.  The
statistics / fused-reduction slabs are compared too; the weight-gradient entries (fp32 atomics) launch to launch within 1e-3 of
the largest element.  YH_RACE_BESIDE=1: every launch next to a busy second stream.  usage: race_screen.py [reps] [key substring]"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from yoloseries_amd import hipk
from yoloseries_amd._lib import YH_ACT_SILU, YH_CONV_DGRAD, check, lib


# kernel families with hand-counted waits / inline-asm transfers (where round 5's two synchronisation errors were: 1 launch in 60 at
# worst): screened with this many launches per entry even when the caller asks for fewer
DEEP_FAMILIES = {"conv_pt_kernel": 40, "conv_halo160_kernel": 40, "conv_wgs_kernel": 40}


def screen(reps=6, sub="", verbose=True, beside=False, deep=None, family=None, burst=1):
    """-> (entries walked, [(key, kernel, differences)]): see the module docstring.  deep: {kernel family: launches per entry}
    overriding `reps` upwards for those families (default DEEP_FAMILIES; {} = none)"""
    deep = DEEP_FAMILIES if deep is None else deep
    dev = torch.device("cuda:0")
    L = lib()
    # beside=True: every screened launch runs next to a stream of memory-bound copies and matmuls on a second stream (the conditions
    # of the two-stream step: other waves on the CUs, a busy vector-memory path), not alone on the chip
    side = torch.cuda.Stream(device=dev) if beside else None
    big = torch.empty(1 << 28, dtype=torch.uint8, device=dev) if beside else None
    mm = (torch.randn(2048, 2048, device=dev), torch.randn(2048, 2048, device=dev)) if beside else None

    def busy():
        if side is None:
            return
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                big[:1 << 27].copy_(big[1 << 27:])
                mm[0] @ mm[1]
    t = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "yoloseries_amd", "tune_defaults.json")))
    keys = [k for k in sorted(t) if k.startswith("conv") and k.split(":")[1] in ("fwd", "dgrad", "eval") and sub in k]
    nan = float("nan")
    bad_entries, fams = [], {}
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)      # noqa: E731
    for ki, key in enumerate(keys):
        f = [int(x) for x in key.split(":")[-1].split(",")]
        (mode, B, Ho, Wo, Hi, Wi, k, stride, pad, N, nseg, C0, ld0s, ups0, C1, ups1, ldo, nsplit, accumulate, stats, res, act, bias, scale, bnr, _z) = f
        tile_k, grid_cap, algo = t[key]
        g = torch.Generator(device=dev).manual_seed(7000 + ki)
        segC, segups = [C0, C1][:nseg], [ups0, ups1][:nseg]
        segs, keep = [], []
        for si in range(nseg):
            h, w_ = Hi >> segups[si], Wi >> segups[si]
            buf = torch.randn(B, h, w_, ld0s if si == 0 else segC[si], generator=g, device=dev).to(torch.bfloat16)
            segs.append(hipk.Slice(buf, 0, segC[si], segups[si])); keep.append(buf)
        Ctot = sum(segC)
        if mode == YH_CONV_DGRAD:
            w = (torch.randn(C0, N, k, k, device=dev, generator=g) / (C0 * k * k) ** 0.5).to(torch.bfloat16).float()
            wp = hipk.pack_weight_dgrad(w)
        else:
            w = (torch.randn(N, Ctot, k, k, device=dev, generator=g) / (Ctot * k * k) ** 0.5).to(torch.bfloat16).float()
            wp = hipk.pack_weight_fwd(w)
        n0 = min(nsplit, N)
        sc = (torch.rand(N, generator=g, device=dev) + 0.5) if scale else None
        sh = (torch.randn(N, generator=g, device=dev) * 0.5) if scale else None
        rs = torch.randn(B, Ho, Wo, n0, generator=g, device=dev).to(torch.bfloat16) if res else None
        acc0 = torch.randn(B, Ho, Wo, ldo, generator=g, device=dev).to(torch.bfloat16) if accumulate else None
        z = ws = None
        if bnr:
            z = torch.randn(B, Ho, Wo, N, generator=g, device=dev).to(torch.bfloat16)
            ws = torch.cat([torch.rand(N, generator=g, device=dev) + 0.5, torch.randn(N, generator=g, device=dev)])
        first, diffs, name = None, [], ""
        r, nrep = -1, reps
        while r + 1 < nrep:
            r += 1
            out0 = acc0.clone() if accumulate else torch.full((B, Ho, Wo, ldo), nan, dtype=torch.bfloat16, device=dev)
            out1 = torch.full((B, Ho, Wo, N - n0 + 8), nan, dtype=torch.bfloat16, device=dev) if n0 < N else None
            d = hipk.conv_desc(segs, mode, B, Ho, Wo, Hi, Wi, k, stride, pad, wp, N, hipk.Slice(out0, 0, n0), nsplit=n0,
                               out1=hipk.Slice(out1, 0, N - n0) if out1 is not None else None, scale=sc, shift=sh,
                               act=YH_ACT_SILU if act else 0, accumulate=accumulate, res=hipk.full(rs) if rs is not None else None)
            d.tile_k, d.grid_cap, d.algo = tile_k, grid_cap, algo
            slab = stt = None
            if stats:
                stt = torch.full((L.yh_conv_stat_blocks(C.byref(d)), 2, wp.shape[0]), nan, device=dev)
                d.stats = stt.data_ptr()
            if bnr:
                d.bnr_z, d.bnr_ldz, d.bnr_C, d.bnr_ws = z.data_ptr(), N, N, ws.data_ptr()
                d.bnr_part = out0.data_ptr()
                slab = torch.full((L.yh_conv_bnr_rows(C.byref(d)), 2, N), nan, device=dev)
                d.bnr_part = slab.data_ptr()
            if r == 0:
                nb = C.create_string_buffer(96); L.yh_conv_kernel_name(C.byref(d), nb, 96); name = nb.value.decode()
                fams[name.split("<")[0]] = fams.get(name.split("<")[0], 0) + 1
                nrep = max(reps, deep.get(name.split("<")[0], 0))
                if family is not None and not name.startswith(family):          # family: only entries of this kernel (name prefix)
                    fams[name.split("<")[0]] -= 1
                    break
            junk = torch.randn(1024, 1024, device=dev) @ torch.randn(1024, 256, device=dev)      # noqa: F841
            busy()
            for _ in range(burst):          # burst > 1: the screen as a LOAD for another process (tools/loss_race_diag.py), results not compared
                check(L.yh_conv_igemm(C.byref(d), st()), key)
            torch.cuda.synchronize()
            cur = [out0[..., :n0].view(torch.int16)] + ([out1[..., :N - n0].view(torch.int16)] if out1 is not None else []) + \
                  ([stt.view(torch.int32)] if stt is not None else []) + ([slab.view(torch.int32)] if slab is not None else [])
            if first is None:
                first = [c.clone() for c in cur]
                if torch.isnan(out0[..., :n0].float()).any():
                    diffs.append(("nan in first launch", int(torch.isnan(out0[..., :n0].float()).sum())))
            else:
                for ti, (a, b) in enumerate(zip(first, cur)):
                    if not torch.equal(a, b):
                        ne = torch.nonzero((a != b).reshape(-1)).flatten()
                        diffs.append((r, f"tensor {ti} of {len(cur)} ({tuple(a.shape)})", int(ne.numel()), ne[:8].tolist()))
                        break
        if diffs:
            bad_entries.append((key, name, diffs))
            if verbose:
                print(f"DIFF {name} {key}: {diffs}", flush=True)
        if verbose and ki % 50 == 49:
            print(f"... {ki + 1} of {len(keys)} entries, {len(bad_entries)} with differences", flush=True)
    # ---- weight gradients: fp32 atomics — not bit-reproducible — so launch to launch within 1e-3 of the largest element
    from yoloseries_amd.engine import Program
    wkeys = [k for k in sorted(t) if k.startswith("wgrad") and sub in k and (family is None or family.startswith("conv_wg"))]
    for ki, key in enumerate(wkeys):
        N, ldg, C0, ld0, ups, Ctot, B, Ho, Wo, Hi, Wi, k, stride, pad = [int(x) for x in key.split(":")[-1].split(",")]
        if key.split(":")[0].endswith("f"):
            continue                               # the stem's fused form needs the BatchNorm operands (tests/test_gpu_tune_table.py)
        splits, tile_k = t[key]
        g = torch.Generator(device=dev).manual_seed(8000 + ki)
        x = torch.randn(B, Hi >> ups, Wi >> ups, ld0, generator=g, device=dev).to(torch.bfloat16)
        gy = (torch.randn(B, Ho, Wo, ldg, generator=g, device=dev) * 0.25).to(torch.bfloat16)
        first, diffs, name = None, [], ""
        r, nrep = -1, reps
        while r + 1 < nrep:
            r += 1
            dw = torch.zeros(N, k * k * Ctot, device=dev)
            d = hipk.wgrad_desc(hipk.Slice(gy, 0, N), N, hipk.Slice(x, 0, C0, ups), Ctot - C0, Ctot, B, Ho, Wo, Hi, Wi, k, stride, pad, dw, splits)
            d.tile_k = tile_k
            if r == 0:
                name = Program._wgrad_name(L, d)
                fams[name.split("<")[0]] = fams.get(name.split("<")[0], 0) + 1
                nrep = max(reps, deep.get(name.split("<")[0], 0))
                if family is not None and not name.startswith(family):          # family: only entries of this kernel (name prefix)
                    fams[name.split("<")[0]] -= 1
                    break
            junk = torch.randn(1024, 1024, device=dev) @ torch.randn(1024, 256, device=dev)      # noqa: F841
            busy()
            for _ in range(burst):
                check(L.yh_conv_wgrad(C.byref(d), st()), key)
            torch.cuda.synchronize()
            if first is None:
                first = dw.clone()
                if not torch.isfinite(dw).all():
                    diffs.append(("non-finite in first launch", int((~torch.isfinite(dw)).sum())))
            else:
                err = (dw - first).abs().max().item()
                if not err <= 1e-3 * first.abs().max().item():
                    diffs.append((r, err))
        if diffs:
            bad_entries.append((key, name, diffs))
            if verbose:
                print(f"DIFF {name} {key}: {diffs}", flush=True)
    if verbose:
        print("kernel families:", dict(sorted(fams.items())))
        print(f"{len(keys)} conv + {len(wkeys)} weight-gradient entries x {reps} launches: {len(bad_entries)} entries with launch-to-launch differences")
    return len(keys) + len(wkeys), bad_entries, fams


if __name__ == "__main__":
    screen(int(sys.argv[1]) if len(sys.argv) > 1 else 6, sys.argv[2] if len(sys.argv) > 2 else "", beside=os.environ.get("YH_RACE_BESIDE") == "1")

