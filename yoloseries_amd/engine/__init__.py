"""Host-side executor of the YOLO conv graph on the HIP kernels (include/yolohip.h).

A model (or a single block used stand-alone) describes itself once to a ``Builder`` as a
list of ops over NHWC bf16 buffers:

  ConvOp  — implicit-GEMM conv over a virtual channel-concat of up to two slices (one may be
            read through a nearest-2x upsample), followed by training-mode BatchNorm + SiLU
            (stats from the conv epilogue, finalize, apply) or, for Detect, a bias only.
            Sibling 1x1 convs that read the same input (C3's cba1/cba2,
            utils/layer_tools.py:165-168) are one GEMM with stacked output channels.
  PoolOp  — SPPF 5x5/s1 max-pool writing into a channel slice of the concat buffer.

torch.cat / nn.Upsample / x.clone() of the reference are never materialised: concat and
upsample are addressing modes of the consumer's loader, residuals are fused into the apply.
``Program`` holds the pre-built kernel descriptors for one input shape and runs
forward (train / eval) and backward; ``ParamPack`` keeps the fp32 master parameters in one
flat arena (the nn.Parameters are views of it, so state_dict/optimizers are unchanged) and
maps them to the packed bf16 weight images and back (packed fp32 grads -> parameter grads)
with one index-gather launch each.
Modules: ``graph`` (planner: Builder, ops, ParamPack, gradient buckets), ``tune`` (launch-parameter tables, per-layer timing),
``executor`` (compiled command arrays), ``forward`` / ``backward`` / ``program`` (the Program of one input shape), ``module``
(autograd node, HipModuleMixin), ``flags`` (environment switches).
"""
from . import flags
from .executor import CompiledCmds
from .flags import BN_EPS_DEFAULT
from .graph import Builder, ConvOp, ParamPack, PoolOp, Ref, TBuf, bn_of, plan_grad_buckets, sppf_chain
from .module import HipModuleMixin
from .program import Program
from .tune import (KEY_CONV, KEY_CONV_C80, KEY_CONV_EVAL, KEY_CONV_H160, KEY_CONV_P3, KEY_CONV_PT, KEY_CONV_S2D, KEY_WGRAD, KEY_WGRAD_WS, TUNE_DEFAULTS_PATH,
                   TUNE_KEY_VERSIONS, tuning_source)

__all__ = ['Builder', 'ConvOp', 'PoolOp', 'Ref', 'TBuf', 'ParamPack', 'Program', 'HipModuleMixin', 'CompiledCmds', 'bn_of', 'plan_grad_buckets',
           'sppf_chain', 'tuning_source', 'flags', 'TUNE_KEY_VERSIONS', 'TUNE_DEFAULTS_PATH']
