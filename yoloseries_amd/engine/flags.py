"""Environment switches of the engine: read once at import into the attributes of this module; the engine reads every switch as
`flags.X` at the point of use, so a test or tool that patches an attribute here changes the next program that is built."""
import os

BN_EPS_DEFAULT = 1e-3

TUNE_ITERS = max(1, int(os.environ.get("YH_TUNE_ITERS", "3")))   # launches timed per candidate (tools/make_tune_defaults.sh: 12)
MERGE_PARTS = os.environ.get("YH_MERGE_PARTS", "1") != "0"   # stacked ConvBnAct layers: one BN+SiLU pass for all parts
# YH_WGRAD_PARTIAL=1: the weight gradients' split-M partial tiles go to a workspace with plain stores and are summed in split
# order by a second kernel (yh_wgrad_desc.partial) instead of fp32 atomics: BIT-REPRODUCIBLE gradients.  Measured on the YOLOv5s
# step: the weight-gradient kernels themselves get 4 % faster (3.67 -> 3.52 ms), the step 1.6 % slower (the 2.5 GB of partial
# tiles are written and read back next to an HBM-bound main chain) — so the atomic form stays the default.
# YH_FUSE_STEM_BWD: the BatchNorm backward apply of a layer without a data gradient (the stem) runs inside its weight gradient's
# operand staging (yh_wgrad_desc.bn_*; the staged gz is the apply pass's gz bit for bit, tests/test_gpu_conv.py): the last pass of
# the backward's critical path and the gz round trip through HBM disappear.  1 (default): where the patch form of the weight
# gradient takes the layer (conv_wgpf_kernel: <= 64 output channels) — measured on the YOLOv5s step 12.70 -> 12.58..12.64 ms (+0.8 %,
# profiles/r03_step_experiments.txt m); 2: also through the im2col form (conv_wgrad_kernel<..., FBN>: 0.55 ms against 0.22 + 0.25 —
# the sigmoid of 210 M elements is hidden behind HBM time in a streaming pass but not between the barriers of a 15-wave-per-CU GEMM;
# the step gets 1 % slower); 0: never.
FUSE_STEM_BWD = int(os.environ.get("YH_FUSE_STEM_BWD", "1"))
HEAD_COLSUM_SIDE = os.environ.get("YH_HEAD_COLSUM_SIDE", "1") != "0"    # bias gradients of the head layers on the weight-gradient stream
SPPF_FUSE = os.environ.get("YH_SPPF_FUSE", "1") != "0"      # FastSPP's three pools in one launch per direction (csrc/sppf.hip)
WG_WS_BYTES = (256 << 20) if os.environ.get("YH_WGRAD_PARTIAL", "0") == "1" else 0
NGZ = int(os.environ.get("YH_GZ_RING", "3"))   # gz buffers the side-stream weight gradients may lag behind by
# YH_SKIP_ALGOS=<n>[,<n>]: leave these kernel families (yh_conv_desc.algo) out of the per-layer timing — A/B runs of a new family on
# one box (use a YH_TUNE_CACHE of its own and YH_TUNE_DEFAULTS=0 for the layers concerned)
SKIP_ALGOS = frozenset(x for x in os.environ.get("YH_SKIP_ALGOS", "").split(",") if x)

# YH_ABL_SKIP=<entry point>[,...|wgrad]: TIMING EXPERIMENTS ONLY (results are wrong) — the named launches are left out of the
# compiled programs, which gives the wall time a step would have if that family were free (profiles/r03_step_ablation.txt)
ABL_SKIP = frozenset(x for x in os.environ.get("YH_ABL_SKIP", "").split(",") if x)

# YH_EXEC=0: launch every kernel of a program from Python (one ctypes call each) instead of replaying the compiled command array
# with one yh_exec call (csrc/exec.hip)
USE_EXEC = os.environ.get("YH_EXEC", "1") != "0"
