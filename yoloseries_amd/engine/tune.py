"""Tuner: the launch-parameter tables (shipped + per-machine) and the per-layer timing of the eligible kernel families
(Program._tune_conv / _tune_wgrad_splits), as a mixin of engine.program.Program."""
import ctypes as C
import json
import os

import torch

from .._lib import YH_CONV_DGRAD, check
from . import flags as _flags

# the table shipped with the package lives beside the package modules
TUNE_DEFAULTS_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tune_defaults.json")


def _tune_cache_path():
    """per-machine timings live in the user's cache directory, not in the package (YH_TUNE_CACHE overrides)"""
    base = os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache")
    return os.environ.get("YH_TUNE_CACHE", os.path.join(base, "yoloseries_amd", "tune_cache.json"))


class _TuneTable:
    """launch parameters per layer shape.  Two layers, never mixed: the shipped table for the BASELINE configurations
    (tune_defaults.json, timed on an MI355X with tools/make_tune_defaults.sh: the same choices on every box, no tuning launches in
    the first steps; read-only) and what THIS machine timed itself for other shapes (a small JSON file, kept across processes).
    A lookup asks the local layer first, then the shipped one; only locally timed keys are ever written back, so a later
    release of tune_defaults.json is not shadowed by a frozen copy of the old one.  YH_TUNE_DEFAULTS=0 ignores the shipped table."""

    def __init__(self):
        self.shipped, self.local, self.dirty = {}, {}, False
        if os.environ.get("YH_TUNE_DEFAULTS", "1") != "0":
            self.shipped = self._read(TUNE_DEFAULTS_PATH)
        self.local = self._read(_tune_cache_path())
        self.hits_shipped = self.hits_local = self.timed = 0

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return dict(json.load(f))
        except (OSError, ValueError):
            return {}

    def __contains__(self, key):
        return key in self.local or key in self.shipped

    def __getitem__(self, key):
        if key in self.local:
            self.hits_local += 1
            return self.local[key]
        self.hits_shipped += 1
        return self.shipped[key]

    def __setitem__(self, key, value):
        self.local[key] = value
        self.timed += 1
        self.dirty = True

    def save(self):
        if not self.dirty:
            return
        self.dirty = False
        path = _tune_cache_path()
        try:
            os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
            tmp = f"{path}.{os.getpid()}.tmp"
            with open(tmp, "w") as f:
                json.dump(self.local, f, indent=0, sort_keys=True)
            os.replace(tmp, path)
        except OSError:
            pass                               # read-only install: tune again next time


def _tune_cache():
    if _tune_cache.data is None:
        _tune_cache.data = _TuneTable()
    return _tune_cache.data


_tune_cache.data = None


def _tune_cache_save():
    if _tune_cache.data is not None:
        _tune_cache.data.save()


def tuning_source():
    """where the launch parameters of this process came from (reported by bench.py)"""
    t = _tune_cache()
    return {"shipped_table": t.hits_shipped, "local_cache": t.hits_local, "timed_now": t.timed}

# wide weight-gradient tilings that also exist with 64-pixel k-steps (yh_wgrad_desc.tile_k = 64): 32-pixel name -> 64-pixel name
_WGRAD_TK64 = {
    "conv_wgrad_kernel<1, 5, 1, 1, 32, 3, true, false>": "conv_wgrad_kernel<1, 5, 1, 1, 64, 3, true, false>",
    "conv_wgrad_kernel<1, 4, 1, 2, 32, 3, true, false>": "conv_wgrad_kernel<1, 4, 1, 2, 64, 2, true, false>",
    "conv_wgrad_kernel<1, 4, 1, 3, 32, 3, false, false>": "conv_wgrad_kernel<1, 4, 1, 3, 64, 2, false, false>",
    "conv_wgrad_kernel<1, 4, 2, 1, 32, 4, false, false>": "conv_wgrad_kernel<1, 4, 2, 1, 64, 2, false, false>",
    "conv_wgrad_kernel<1, 4, 2, 2, 32, 3, false, false>": "conv_wgrad_kernel<1, 4, 2, 2, 64, 2, false, false>",
}


# version prefixes of the tuning-table keys: bumped when the candidates or the meaning of a tuned value change, so that stale
# entries of a shipped / cached table are not applied.  Stride-2 data gradients carry their own version (conv_dg2_kernel, algo 7,
# joined their candidates in round 3), and so do weight gradients that leave through the partial-tile workspace.
KEY_CONV, KEY_CONV_S2D, KEY_CONV_P3, KEY_CONV_EVAL, KEY_WGRAD, KEY_WGRAD_WS = "conv6", "conv7", "conv8", "conv9", "wgrad10", "wgrad8"
KEY_CONV_C80 = "conv10"        # inference 3x3 layers with 80 -> 160 channels: conv_c80_kernel (algo 12) joined their candidates in round 4
KEY_CONV_PT = "conv11"         # 1x1 layers conv_pt_kernel (algo 13) takes: training with 128 / 256 / 512 input channels, inference with 320 (round 5)
KEY_CONV_H160 = "conv14"       # inference 3x3 / stride-1 layers with N a multiple of 160: re-timed against the FINAL conv_halo160_kernel of round 5 (16 x 16 tiles, pipelined sub-steps): it now takes every one of them, also the 640-channel layers the conv12 entries had left on conv_halo_kernel
TUNE_KEY_VERSIONS = frozenset((KEY_CONV, KEY_CONV_S2D, KEY_CONV_P3, KEY_CONV_EVAL, KEY_CONV_C80, KEY_CONV_PT, KEY_CONV_H160, KEY_WGRAD, KEY_WGRAD_WS, KEY_WGRAD + "f", KEY_WGRAD_WS + "f"))


class TunerMixin:
    """timed choice of kernel family / launch parameters per layer shape (mixed into Program)"""

    def _tune_conv(self, d, kind, name, stats_ok=False):
        """Launch parameters of one conv / dgrad launch — kernel family (register-staged conv_v2 or LDS-DMA conv_v3 with one
        of its tiles), k-step width, cap on persistent blocks — timed once when the program is built: the best setting
        differs per layer shape by 5-40 % (YH_CONV_TUNE=0: library defaults).  Results never change (identical math);
        only the number of BatchNorm partial-sum rows follows the grid."""
        if os.environ.get("YH_CONV_TUNE", "1") == "0":
            return
        small3 = d.KH == 3 and d.stride == 1 and d.nseg == 1 and d.seg[0].C <= 128 and d.N <= 128 and kind != 'eval'
        c80 = kind == 'eval' and d.KH == 3 and d.nseg == 1 and d.seg[0].C == 80 and d.N == 160
        h160 = kind == 'eval' and d.KH == 3 and d.stride == 1 and d.nseg == 1 and d.N % 160 == 0 and d.seg[0].C >= 64 and d.seg[0].C % 32 == 0
        ctot = d.seg[0].C + (d.seg[1].C if d.nseg > 1 else 0)
        pt = kind != 'eval' and d.KH == 1 and d.stride == 1 and ctot in (128, 256, 512) and (d.nseg == 1 or d.seg[0].C == d.seg[1].C)
        pte = kind == 'eval' and d.KH == 1 and d.stride == 1 and d.nseg == 1 and ctot == 320 and d.nsplit >= d.N     # conv_pt_kernel's inference form
        key = f"{KEY_CONV_S2D if d.mode == YH_CONV_DGRAD and d.stride == 2 else (KEY_CONV_P3 if small3 else ((KEY_CONV_C80 if c80 else (KEY_CONV_H160 if h160 else (KEY_CONV_PT if pte else KEY_CONV_EVAL))) if kind == 'eval' else (KEY_CONV_PT if pt else KEY_CONV)))}:{kind}:" + ",".join(str(int(v)) for v in (
            d.mode, d.B, d.Ho, d.Wo, d.Hi, d.Wi, d.KH, d.stride, d.pad, d.N, d.nseg, d.seg[0].C, d.seg[0].ld, d.seg[0].ups,
            d.seg[1].C if d.nseg > 1 else 0, d.seg[1].ups if d.nseg > 1 else 0, d.ld0, d.nsplit, d.accumulate, int(bool(d.stats or stats_ok)),
            int(bool(d.res)), d.act, int(bool(d.bias)), int(bool(d.scale)), int(bool(d.bnr_part)), 0))
        cache = _tune_cache()
        if key in cache:
            d.tile_k, d.grid_cap, d.algo = (int(v) for v in cache[key])
            return
        L = self.L
        saved = (d.seg[0].ptr, d.stats)
        if not d.seg[0].ptr:
            d.seg[0].ptr = self.gy_scratch.data_ptr()
        # candidates: (algo, tile_k, grid_cap)
        d.tile_k = d.grid_cap = 0
        d.algo = 1
        base = L.yh_conv_stat_blocks(C.byref(d))
        cands = []
        tks = (0, 32) if all(d.seg[i].C % 64 == 0 for i in range(d.nseg)) and d.N > 64 else (0,)
        for tk in tks:
            for cap in (0, 2 * base):
                d.tile_k, d.grid_cap = tk, cap
                if cap and L.yh_conv_stat_blocks(C.byref(d)) == base:
                    continue                       # fewer tiles than blocks: the cap changes nothing
                cands.append((1, tk, cap))
        d.tile_k = d.grid_cap = 0
        if os.environ.get("YH_CONV_V3", "1") != "0":
            for algo in (2, 3, 4, 5, 6, 7, 8, 9, 10, 12, 13, 14):
                if str(algo) in _flags.SKIP_ALGOS:
                    continue
                d.algo = algo
                kn = self._kernel_name(d)
                if ("conv_v3" in kn and algo < 5) or ("conv_halo_kernel" in kn and algo == 5) or ("conv_halo160" in kn and algo == 6) or \
                        ("conv_dg2" in kn and algo == 7) or ("conv_p3" in kn and algo == 8) or ("conv_h80" in kn and algo == 9) or ("conv_pw" in kn and algo == 10) or ("conv_c80" in kn and algo == 12) or \
                        ("conv_pt" in kn and algo == 13) or ("conv_v3_kernel<256, 256" in kn and algo == 14):
                    cands.append((algo, 0, 0))
                    if algo < 5 and kn.endswith(", true>") and all(d.seg[i].C % 32 == 0 for i in range(d.nseg)):
                        cands.append((algo, 32, 0))    # ragged last channel block: 32-channel steps instead of 64 + tail
        rows_max, bnr_max = 1, 1
        for algo, tk, cap in cands:
            d.algo, d.tile_k, d.grid_cap = algo, tk, cap
            rows_max = max(rows_max, L.yh_conv_stat_blocks(C.byref(d)))
            if d.bnr_part:
                bnr_max = max(bnr_max, L.yh_conv_bnr_rows(C.byref(d)))
        tmp_stats = None
        if stats_ok:
            tmp_stats = torch.zeros(rows_max + 8, 2, d.Npad, dtype=torch.float32, device=self.dev)
            d.stats = tmp_stats.data_ptr()
        saved_part, tmp_part = d.bnr_part, None
        if d.bnr_part:                   # a slab big enough for every grid tried below
            tmp_part = torch.zeros((bnr_max + 8) * 2 * d.N, dtype=torch.float32, device=self.dev)
            d.bnr_part = tmp_part.data_ptr()
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        best, best_ms = (0, 0, 1), None
        for algo, tk, cap in cands:
            d.algo, d.tile_k, d.grid_cap = algo, tk, cap
            check(L.yh_conv_igemm(C.byref(d), st), f"yh_conv_igemm tune [{name}]")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(_flags.TUNE_ITERS):
                L.yh_conv_igemm(C.byref(d), st)
            e1.record()
            e1.synchronize()
            ms = e0.elapsed_time(e1)
            if best_ms is None or ms < best_ms * (0.97 if _flags.TUNE_ITERS < 8 else 0.99):   # keep the earlier candidate unless clearly better
                best, best_ms = (tk, cap, algo), ms
        d.tile_k, d.grid_cap, d.algo = best
        d.seg[0].ptr, d.stats = saved
        d.bnr_part = saved_part
        cache[key] = [int(best[0]), int(best[1]), int(best[2])]

    def _kernel_name(self, d):
        """instantiation yh_conv_igemm launches for descriptor d, spelled as rocprofv3 prints it"""
        buf = C.create_string_buffer(96)
        saved = d.seg[0].ptr
        if not saved:                      # head gradient pointer is filled in at run time
            d.seg[0].ptr = self.gy_scratch.data_ptr()
        rc = self.L.yh_conv_kernel_name(C.byref(d), buf, 96)
        d.seg[0].ptr = saved
        check(rc, "yh_conv_kernel_name")
        return buf.value.decode()

    def _tune_wgrad_splits(self, wd, M, ntile, op):
        """Split-M factor (and, for the wide 64-row tilings, the pixels per k-step) of one weight-gradient launch.  The best
        total block count depends on the tile configuration's residency and on how the atomics of the epilogue amortise
        (measured 256..1024 blocks, up to 1.6x apart), so it is timed once per layer when the backward program is built
        (YH_WGRAD_TUNE=0: fixed 512-block rule).  Sets wd.tile_k, returns the split factor."""
        Kseg = wd.KH * wd.KW * wd.seg.C

        def splits_for(total, tk=0):
            nt = self.L.yh_conv_wgrad_tiles2(wd.N, Kseg, tk) if tk == 128 else ntile
            return max(1, min((M + 255) // 256, (total + nt - 1) // nt))
        if os.environ.get("YH_WGRAD_TUNE", "1") == "0":
            return splits_for(512)
        key = f"{KEY_WGRAD_WS if wd.partial else KEY_WGRAD}{'f' if wd.bn_z else ''}:" + ",".join(str(int(v)) for v in (wd.N, wd.ldg, wd.seg.C, wd.seg.ld, wd.seg.ups, wd.Ctot, wd.B, wd.Ho, wd.Wo,
                                                          wd.Hi, wd.Wi, wd.KH, wd.stride, wd.pad))
        cache = _tune_cache()
        if key in cache:
            sp, wd.tile_k = (int(v) for v in cache[key])
            return sp
        gy_saved = wd.gy
        if not wd.gy:                      # head gradient arrives at run time: time against the scratch buffer
            if self.gy_scratch.numel() < M * wd.ldg:
                return splits_for(512)
            wd.gy = self.gy_scratch.data_ptr()
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        name = self.L.yh_conv_wgrad_kernel_name(wd.N, wd.KH * wd.KW * wd.seg.C).decode()
        tks = (0, 64) if name in _WGRAD_TK64 else (0,)
        if name.startswith("conv_wgrad_kernel<4, 2, 1, 2, 64"):
            tks = tks + (32, 35)            # the general tiling with 32-pixel k-steps (two blocks per CU): 8 waves of 32 x 64 / 4 of 64 x 64
        if 128 <= Kseg <= 384 and wd.N > 32 and not wd.bn_z:
            tks = tks + (128,)              # the general 128-column tiling on a layer that defaults to a wide one
        best, best_ms = None, None
        wd.tile_k = 40
        if not wd.partial and self.L.yh_conv_wgrad_patch_ok(C.byref(wd)):
            tks = tks + (40,)               # patch form (conv_wgp_kernel): the input patch of a pixel region staged once in LDS
        wtiles = 0 if (wd.partial or wd.bn_z or os.environ.get("YH_WGRAD_WAVE", "1") == "0") else self.L.yh_conv_wgrad_wave_tiles(C.byref(wd))
        if wtiles > 0:
            tks = tks + (129,)              # wave-private 128 x 128 tiles + stream-K (conv_wgs_kernel): `splits` = workgroups, one per CU
        for tk in tks:
            wd.tile_k = tk
            if tk == 129:                   # an exact tiles x splits grid where it fills the chip, else 256 workgroups dealt (tile, 32 pixels) units
                # Workgroups (= CUs: the form holds a whole CU) a weight gradient may take.  Alone on the chip 256 is fastest; in the
                # two-stream backward the main chain runs beside it, and its short latency-bound kernels (finalize launches, small
                # layers) wait for a CU while a weight gradient holds all of them: the YOLOv5s step is shortest when the weight
                # gradients leave a quarter of the CUs alone (12.00 -> 11.90 ms), the YOLOv5l step — long kernels on both streams —
                # when its big layers take the whole chip (43.36 -> 42.82 ms): layers under 60 GFLOP get 192, the others 256
                # (profiles/r04_step_experiments.txt d).  Half of the budget is timed too: on the small layers the atomics (one
                # partial tile per workgroup) dominate.
                wflops = 2.0 * M * wd.N * wd.KH * wd.KW * wd.seg.C
                gmax = int(os.environ.get("YH_WGS_G", "256" if wflops >= 60e9 else "192"))
                sps = set()
                for g in (gmax, gmax // 2):
                    sps |= {g} | ({wtiles * (g // wtiles)} if wtiles <= g else set())
                sps = sorted(sps)
            else:
                sps = [1024] if tk == 40 else sorted({splits_for(t, tk) for t in (256, 512, 768, 1024, 1536)})
            for sp in sps:
                wd.splits = sp
                if wd.partial and self.L.yh_conv_wgrad_ws_bytes(C.byref(wd)) > wd.partial_bytes:
                    continue                   # more partial tiles than the workspace holds
                check(self.L.yh_conv_wgrad(C.byref(wd), st), f"yh_conv_wgrad tune [{op.name}]")
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(_flags.TUNE_ITERS):
                    self.L.yh_conv_wgrad(C.byref(wd), st)
                e1.record()
                e1.synchronize()
                ms = e0.elapsed_time(e1)
                if best_ms is None or ms < best_ms:
                    best, best_ms = (sp, tk), ms
        wd.gy = gy_saved
        wd.tile_k = best[1]
        self.wgrad_tuned[(op.name, wd.coff_k)] = (best[0], best_ms / _flags.TUNE_ITERS)
        cache[key] = [int(best[0]), int(best[1])]
        return best[0]

    @staticmethod
    def _wgrad_name(L, wd):
        """instantiation yh_conv_wgrad launches for this descriptor, profiler spelling (64-pixel k-steps on the wide tilings:
        csrc/conv_wgrad.hip, yh_conv_wgrad)"""
        name = L.yh_conv_wgrad_kernel_name2(wd.N, wd.KH * wd.KW * wd.seg.C, wd.tile_k).decode()
        if wd.tile_k == 64:
            name = _WGRAD_TK64.get(name, name)
        if wd.bn_z:                        # last template argument: BatchNorm backward fused into the operand loader
            name = name[:-len(", false>")] + ", true>"
        if wd.tile_k == 129 and L.yh_conv_wgrad_wave_tiles(C.byref(wd)) > 0:
            return L.yh_conv_wgrad_wave_name(C.byref(wd)).decode()
        if wd.tile_k == 40 and L.yh_conv_wgrad_patch_ok(C.byref(wd)):
            buf = C.create_string_buffer(64)
            L.yh_conv_wgrad_patch_name(C.byref(wd), buf, 64)
            name = buf.value.decode() or name
        return name
