#!/usr/bin/env python3
"""Fill the R4_* placeholders of DESIGN.md / README.md from profiles/r04_bench_*.json (run after tools/profile_round.sh r04 and copying its
outputs to profiles/): the numbers quoted in the documents are the committed profile set's.   usage: tools/fill_design_r04.py [file ...]"""
import json, sys, re

def load(name):
    return json.loads(open(f"profiles/{name}.json").read().strip().split("\n")[-1])

s5, yx, l5, inf = load("r04_bench_default"), load("r04_bench_yolox"), load("r04_bench_v5l"), load("r04_bench_infer_v5x_1280_b128")
def g(d, grp, key="ms_per_step"):
    return d["roofline"]["groups"][grp][key]
def fmt(v, nd=0):
    return f"{v:,.{nd}f}".replace(",", " ")
conv = inf["roofline"]["conv"]
vals = {
    "R4_V5S_MS": f"{s5['ms_per_step']:.2f}", "R4_V5S_TF": f"{s5['roofline']['conv_kernels_tflops']:.0f}", "R4_V5S_MU": f"{100 * s5['roofline']['mfma_util_step']:.1f} %",
    "R4_V5S_HBM": f"{s5['roofline']['hbm_bytes_per_step'] / 1e9:.1f}", "R4_V5S_GC": f"{g(s5, 'conv'):.2f}", "R4_V5S_GW": f"{g(s5, 'wgrad'):.2f}",
    "R4_V5S_GB": f"{g(s5, 'bn_silu'):.2f}", "R4_V5S_GF": f"{g(s5, 'finalize'):.2f}", "R4_V5S_GO": f"{g(s5, 'other'):.2f}", "R4_V5S": fmt(s5["value"]),
    "R4_YX_MS": f"{yx['ms_per_step']:.2f}", "R4_YX_TF": f"{yx['roofline']['conv_kernels_tflops']:.0f}", "R4_YX_MU": f"{100 * yx['roofline']['mfma_util_step']:.1f} %",
    "R4_YX_HBM": f"{yx['roofline']['hbm_bytes_per_step'] / 1e9:.1f}", "R4_YX_GW": f"{g(yx, 'wgrad'):.2f}", "R4_YX": fmt(yx["value"]),
    "R4_V5L_MS": f"{l5['ms_per_step']:.1f}", "R4_V5L_TF": f"{l5['roofline']['conv_kernels_tflops']:.0f}", "R4_V5L_MU": f"{100 * l5['roofline']['mfma_util_step']:.1f} %",
    "R4_V5L_HBM": f"{l5['roofline']['hbm_bytes_per_step'] / 1e9:.0f}", "R4_V5L_GCTF": f"{g(l5, 'conv', 'tflops'):.0f}", "R4_V5L_GC": f"{g(l5, 'conv'):.1f}",
    "R4_V5L_GWTF": f"{g(l5, 'wgrad', 'tflops'):.0f}", "R4_V5L_GW": f"{g(l5, 'wgrad'):.1f}", "R4_V5L_GB": f"{g(l5, 'bn_silu'):.1f}", "R4_V5L_GF": f"{g(l5, 'finalize'):.1f}",
    "R4_V5L": fmt(l5["value"]),
    "R4_INF_MS": f"{inf['ms_per_step']:.0f}", "R4_INF_TF": f"{inf['roofline']['conv_kernels_tflops']:.0f}", "R4_INF_MU": f"{100 * inf['roofline']['mfma_util_step']:.1f} %",
    "R4_INF_HBM": f"{inf['roofline']['hbm_bytes_per_step'] / 1e9:.0f}", "R4_INF_HALO": f"{conv['achieved']:.0f}",
    "R4_INF_HT": f"{conv['traffic'] / conv['bytes_per_launch']:.2f}" if conv.get("traffic") else "n/a", "R4_INF": fmt(inf["value"]),
}
for path in (sys.argv[1:] or ["DESIGN.md"]):
    t = open(path).read()
    for k in sorted(vals, key=len, reverse=True):
        t = t.replace(k, vals[k])
    open(path, "w").write(t)
    left = sorted(set(re.findall(r"R4_[A-Z0-9_]+", t)))
    print(path, "placeholders left:", left)
