"""Executor: a pass of the engine as a yh_cmd array (include/yolohip.h) replayed by one yh_exec call per segment."""
import ctypes as C
import struct


from .._lib import Cmd, YH_CMD_EVENT_RECORD, YH_CMD_SLOTS, YoloHipError, check


def _slot(v):
    """a command argument widened to the 8-byte slot yh_exec expects"""
    if v is None:
        return 0
    if isinstance(v, float):
        return struct.unpack("<Q", struct.pack("<d", v))[0]
    if isinstance(v, (C.Structure, C.Array)):
        return C.addressof(v)
    return int(v) & 0xFFFFFFFFFFFFFFFF


class CompiledCmds:
    """a command list as a yh_cmd array (include/yolohip.h): built once per program, replayed with one call per segment"""

    def __init__(self, L, capacity):
        self.L = L
        self.arr = (Cmd * max(capacity, 1))()
        self.n = 0
        self.names = []
        self.source = None              # the Python command list this array was compiled from

    def call(self, fn, args, stream=0, label=""):
        nargs = C.c_int32(0)
        op = self.L.yh_exec_op(fn.__name__.encode(), C.byref(nargs))
        if op < 0 or nargs.value != len(args) + 1 or nargs.value > YH_CMD_SLOTS:
            raise YoloHipError(f"{fn.__name__} [{label}] cannot be compiled into a program ({len(args)} arguments)")
        c = self.arr[self.n]
        c.op, c.nslots, c.stream = op, nargs.value, stream
        for i, v in enumerate(args):
            c.slots[i] = _slot(v)
        self.names.append(f"{fn.__name__} [{label}]")
        self.n += 1
        return self.n - 1

    def event(self, kind, handle, stream):
        c = self.arr[self.n]
        c.op, c.nslots, c.stream = kind, 1, stream
        c.slots[0] = int(handle)
        self.names.append("event record" if kind == YH_CMD_EVENT_RECORD else "stream wait")
        self.n += 1

    def run(self, streams, lo=0, hi=None):
        hi = self.n if hi is None else hi
        if hi <= lo:
            return
        failed = C.c_int32(-1)
        arr = (C.c_void_p * len(streams))(*streams)
        rc = self.L.yh_exec(C.cast(C.byref(self.arr, lo * C.sizeof(Cmd)), C.POINTER(Cmd)), hi - lo, arr, len(streams), C.byref(failed))
        if rc != 0:
            check(rc, self.names[lo + failed.value] if failed.value >= 0 else "yh_exec")
