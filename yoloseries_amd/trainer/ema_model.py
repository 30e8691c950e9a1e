"""ExponentialMovingAverageModel — mirror of trainer/ema_model.py:7-28.
decay = decay_ratio * (1 - exp(-n/2000)); every floating tensor of the state_dict follows
e = d*e + (1-d)*p.  When the source model keeps its parameters / BatchNorm statistics in the
engine's flat arenas, the whole update is two HIP launches (parameters, float buffers)."""
import math
from copy import deepcopy

import torch

__all__ = ['ExponentialMovingAverageModel']


def _unwrap(model):
    return model.module if hasattr(model, 'module') and isinstance(model.module, torch.nn.Module) else model


class ExponentialMovingAverageModel:

    def __init__(self, model, decay_ratio=0.9999, update_num=0):
        self.ema = deepcopy(_unwrap(model)).eval()
        self.update_num = update_num
        self.decay_ratio = decay_ratio
        self.get_decay_weight = lambda x: decay_ratio * (1 - math.exp(-x / 2000))
        self._dev_state = None          # (counter int64[1], decay float32[1]) on the device + the count it holds
        for parm in self.ema.parameters():
            parm.requires_grad_(False)
        self._flat = None
        self._fast = None               # (source pack, destination arenas) validated by an earlier update

    def _flat_views(self, src_pack):
        """(re)alias the EMA module's parameters / float buffers onto two flat tensors laid out like the source arenas"""
        ps = list(self.ema.parameters())
        ok = self._flat is not None and ps[0].data_ptr() == self._flat[0].data_ptr() and self._flat[0].numel() == src_pack.n
        if not ok:
            fp = torch.empty(src_pack.n, dtype=torch.float32, device=src_pack.device)
            o = 0
            for p in ps:
                fp[o:o + p.numel()].copy_(p.data.reshape(-1))
                p.data = fp[o:o + p.numel()].view(p.shape)
                o += p.numel()
            fb = torch.empty(max(src_pack.nbuf, 1), dtype=torch.float32, device=src_pack.device)
            o = 0
            for b in self.ema.buffers():
                if b.dtype == torch.float32:
                    fb[o:o + b.numel()].copy_(b.data.reshape(-1))
                    b.data = fb[o:o + b.numel()].view(b.shape)
                    o += b.numel()
            self._flat = (fp, fb)
            if hasattr(self.ema, "_yh_reset"):
                self.ema._yh_reset()
        return self._flat

    def graph_pre_replay(self):
        """host-side bookkeeping of one update that runs as a graph replay (the device counter advances inside the graph)"""
        self.update_num += 1
        if self._dev_state is not None:
            self._dev_state[2] = self.update_num

    def graph_snapshot(self):
        return self.update_num

    def graph_restore(self, snap):
        """a capture failed after update() had counted an update whose kernels never ran: back to the count the device holds;
        the device counter / decay pair is rebuilt by the next eager update"""
        self.update_num = snap
        self._dev_state = None
        self._fast = None

    def _dst_unchanged(self, fast):
        """the EMA module's parameters still live where the last full update found them: in its own engine arenas (it has been
        evaluated) or in the flat views of _flat_views() (it has not — and still has no engine state of its own)"""
        est = self.ema.__dict__.get('_yh')
        epack = est['pack'] if est else None
        if fast[3] is not None:
            return epack is fast[3] and epack.still_valid()
        return epack is None and next(self.ema.parameters()).data_ptr() == fast[1].data_ptr()

    def update(self, model):
        with torch.no_grad():
            self.update_num += 1
            decay_weight = self.get_decay_weight(self.update_num)
            src = _unwrap(model)
            st = src.__dict__.get('_yh')
            pack = st['pack'] if st else None
            from .. import hipk
            fast = self._fast
            # every 64th update takes the full path below (valid_for() walks every parameter of both modules: a re-created or
            # replaced MIDDLE parameter, which the first / last pointer checks of still_valid() cannot see, ends the fast path there)
            if fast is not None and (self.update_num & 63) != 0 and fast[0] is pack and pack.still_valid() and self._dev_state is not None and \
                    self._dev_state[2] == self.update_num - 1 and self._dst_unchanged(fast):
                # steady state of a training loop: same arenas as last time — three launches and no walk over the 177 parameters
                # (the full validity checks below cost ~0.2 ms of host time per step, a visible gap in a profiler trace)
                cnt, dec = self._dev_state[0], self._dev_state[1]
                hipk.ema_advance(cnt, dec, self.decay_ratio, 2000.0)
                self._dev_state[2] = self.update_num
                hipk.ema_update_dev(fast[1], pack.flat, dec)
                if pack.nbuf:
                    hipk.ema_update_dev(fast[2], pack.fbuf, dec)
                return
            if pack is not None and pack.valid_for(src):
                # the decay of this update is produced on the device from a device-resident update counter (same formula,
                # evaluated in double): the launch takes no host scalar, so a captured step replays correctly
                if self._dev_state is None or self._dev_state[2] != self.update_num - 1 or self._dev_state[0].device != pack.device:
                    if torch.cuda.is_current_stream_capturing():
                        raise RuntimeError("ExponentialMovingAverageModel: update counter changed outside update(); run one "
                                           "eager update before capturing")
                    cnt = torch.tensor([self.update_num - 1], dtype=torch.int64).to(pack.device)
                    dec = torch.zeros(1, dtype=torch.float32, device=pack.device)
                    self._dev_state = [cnt, dec, self.update_num - 1]
                cnt, dec = self._dev_state[0], self._dev_state[1]
                hipk.ema_advance(cnt, dec, self.decay_ratio, 2000.0)
                self._dev_state[2] = self.update_num
                def _ema(dst, srcv):
                    hipk.ema_update_dev(dst, srcv, dec)
                est = self.ema.__dict__.get('_yh')
                epack = est['pack'] if est else None
                if epack is not None and epack.valid_for(self.ema) and epack.n == pack.n and epack.nbuf == pack.nbuf:
                    # the EMA module has been evaluated: its parameters live in its own engine arenas (same layout as the
                    # source's) — update those in place and keep its programs / activation buffers / tuned descriptors
                    _ema(epack.flat, pack.flat)
                    if pack.nbuf:
                        _ema(epack.fbuf, pack.fbuf)
                    self._fast = (pack, epack.flat, epack.fbuf, epack)
                    return
                fp, fb = self._flat_views(pack)
                _ema(fp, pack.flat)
                if pack.nbuf:
                    _ema(fb, pack.fbuf)
                self._fast = (pack, fp, fb, None)
                # integer buffers (num_batches_tracked) are not averaged by the reference either
                return
            state = src.state_dict()
            for k, v in self.ema.state_dict().items():
                if v.dtype.is_floating_point:
                    v *= decay_weight
                    v += (1. - decay_weight) * state[k].detach()
