"""The shipped library holds no packed-fp32 instruction that takes its LOW result from the HIGH dword of a second / third source
(`v_pk_{add,mul,fma}_f32 ... op_sel:[x,1]`).  On the MI355X boxes of this pool that form returns a wrong low half while another
wave of the SIMD executes MFMA (tools/pk_victim_probe.py: 0 wrong of 10^13 alone, 5 % of the threads beside
`v_mfma_f32_16x16x32_bf16`) — it is how the YOLOv5 loss backward went wrong beside a second training process.
The SLP vectorizer forms it from (x, y) / (w, h) arithmetic; the exact sources compile without the vectorizers (csrc/Makefile EXACT),
the bf16 activation path never had it.  This test disassembles every gfx950 code object of libyolohip.so (no GPU needed)."""
import os
import re
import struct
import subprocess

import pytest

from yoloseries_amd import _lib

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def _code_objects(path):
    """the gfx950 ELF images of a HIP fat binary (clang offload bundles inside the shared library)"""
    d = open(path, "rb").read()
    magic, pos, out = b"__CLANG_OFFLOAD_BUNDLE__", 0, []
    while True:
        i = d.find(magic, pos)
        if i < 0:
            return out
        n = struct.unpack_from("<Q", d, i + 24)[0]
        off = i + 32
        for _ in range(n):
            eo, es, ts = struct.unpack_from("<QQQ", d, off)
            off += 24
            triple = d[off:off + ts].decode(errors="replace")
            off += ts
            if "gfx950" in triple and es > 0:
                out.append(d[i + eo:i + eo + es])
        pos = i + 24


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump of the ROCm toolchain not found")
def test_no_packed_fp32_form_reads_the_high_dword_into_the_low_lane(tmp_path):
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    objs = _code_objects(_lib.LIB_PATH)
    assert len(objs) >= 10, "expected one gfx950 code object per HIP source, found %d" % len(objs)
    packed, bad = 0, []
    for k, blob in enumerate(objs):
        f = tmp_path / ("co_%d.elf" % k)
        f.write_bytes(blob)
        dis = subprocess.run([OBJDUMP, "-d", str(f)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, check=True).stdout
        for line in dis.splitlines():
            m = re.search(r"\bv_pk_(add|mul|fma)_f32\b.*", line)
            if not m:
                continue
            packed += 1
            sel = re.search(r"op_sel:\[([01](?:,[01])+)\]", m.group(0))
            if sel and "1" in sel.group(1).split(",")[1:]:
                bad.append(m.group(0).split("//")[0].strip())
    assert packed > 1000, "the disassembly found %d packed fp32 instructions: the scan does not see the conv epilogues" % packed
    assert not bad, "%d packed fp32 instructions with op_sel on a second / third source (DESIGN §5 (l)): %s" % (len(bad), bad[:5])
