#!/usr/bin/env python3
"""Which packed-fp32 instruction form goes wrong beside which kernels?  tools/micro/pk_victim.hip's victim kernel
(one packed form per launch, checked against scalar instructions in the same thread) on the main stream while a second stream of
this process runs
   <family>     launches of ONE kernel family of the shipped table (tools/race_screen.py in bursts, from a thread), e.g. conv_wgs_kernel
   micro:all    the micro triggers of pk_victim.hip one by one (valu, accvgpr, mfma32_vgpr, mfma32_agpr, mfma16_vgpr, ds_read_tr),
                victims: plain v_pk_mul_f32 and the two op_sel:[0,1] forms
   none         nothing
usage: pk_victim_probe.py [family|micro:<kinds>|none] [seconds per form]     (about a minute on one MI355X)"""
import ctypes as C
import os
import subprocess
import sys
import threading
import time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import torch

FORMS = [(0, "v_pk_mul_f32"), (1, "v_pk_mul_f32 neg_lo neg_hi"), (4, "v_pk_fma_f32"), (5, "v_pk_mov_b32 op_sel:[1,0]"),
         (6, "v_cmp / v_cndmask -> v_pk_mul_f32 neg"),
         # every packed form of the -O3 build of v5_pos_bwd_kernel (X, Y: VGPR pairs, S: an SGPR pair)
         (100, "v_pk_add_f32 D, X, Y neg_lo:[0,1] neg_hi:[0,1]"),
         (101, "v_pk_add_f32 D, X, Y"),
         (102, "v_pk_mul_f32 D, X, Y"),
         (103, "v_pk_mul_f32 D, X, Y op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]"),
         (104, "v_pk_add_f32 D, X, Y op_sel:[1,0] op_sel_hi:[0,1]"),
         (105, "v_pk_add_f32 D, X, Y op_sel:[0,1] op_sel_hi:[1,0]"),
         (106, "v_pk_mul_f32 D, X, Y op_sel:[0,1] op_sel_hi:[1,0]"),
         (107, "v_pk_mul_f32 D, X, Y op_sel_hi:[0,1]"),
         (108, "v_pk_mul_f32 D, X, 0 op_sel_hi:[1,0]"),
         (109, "v_pk_add_f32 D, X, Y op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]"),
         (110, "v_pk_mul_f32 D, X, 0.5 op_sel_hi:[1,0]"),
         (111, "v_pk_add_f32 D, X, Y op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]"),
         (112, "v_pk_mul_f32 D, X, Y op_sel_hi:[1,0]"),
         (113, "v_pk_mul_f32 D, X, S op_sel_hi:[0,1]"),
         (114, "v_pk_add_f32 D, X, S neg_lo:[1,0] neg_hi:[1,0]"),
         (115, "v_pk_add_f32 D, X, S"),
         (116, "v_pk_add_f32 D, X, 1.0 op_sel_hi:[1,0]"),
         (117, "v_pk_mul_f32 D, X, Y neg_lo:[0,1] neg_hi:[0,1]"),
         (118, "v_pk_add_f32 D, X, Y op_sel_hi:[0,1]"),
         (119, "v_pk_add_f32 D, X, Y op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]"),
         (120, "v_pk_add_f32 D, X, 1.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]"),
         (121, "v_pk_add_f32 D, X, -0.5 op_sel_hi:[1,0]"),
         # more selectors on the second / third source
         (200, "v_pk_mul_f32 D, X, Y op_sel:[0,1]"), (201, "v_pk_add_f32 D, X, Y op_sel:[1,1] op_sel_hi:[0,0]"),
         (202, "v_pk_fma_f32 D, X, Y, Z op_sel:[0,1,0] op_sel_hi:[1,0,1]"), (203, "v_pk_fma_f32 D, X, Y, Z op_sel:[0,0,1] op_sel_hi:[1,1,0]"),
         (204, "v_pk_mov_b32 D, X, Y op_sel:[0,1]"), (205, "v_pk_mov_b32 D, X, Y op_sel:[1,1]"),
         ]
VICTIMS_MICRO = tuple(int(v) for v in os.environ.get("PK_VICTIMS", "0,105,106").split(","))


def main():
    family = sys.argv[1] if len(sys.argv) > 1 else "conv_wgs_kernel"
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
    so = os.path.join(HERE, "micro", "libpk_victim.so")
    if not os.path.exists(so):
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-fno-slp-vectorize", "-fno-vectorize", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so,
                        os.path.join(HERE, "micro", "pk_victim.hip")], check=True)
    L = C.CDLL(so)
    L.pk_victim_launch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    dev = torch.device("cuda", 0)
    inp = (torch.rand(1 << 16, device=dev) * 4 - 2)
    stop = []
    if family.startswith("micro:"):          # a micro trigger of pk_victim.hip on a second stream, victim = the op_sel:[0,1] forms only
        kinds = {"valu": 0, "accvgpr": 1, "mfma32_vgpr": 2, "mfma32_agpr": 3, "mfma16_vgpr": 4, "ds_read_tr": 5, "mfma16_agpr": 6,
                 "mfma32x8_vgpr": 7, "mfma16x16_vgpr": 8, "mfma32_vgpr_x4": 9}
        L.pk_trigger_launch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        side = torch.cuda.Stream(dev)
        for kname in (family.split(":", 1)[1].split(",") if family != "micro:all" else list(kinds)):
            for form, name in [f for f in FORMS if f[0] in VICTIMS_MICRO]:
                mism = torch.zeros(4, dtype=torch.int64, device=dev)
                first = torch.zeros(1, dtype=torch.int32, device=dev)
                t0, launches = time.time(), 0
                while time.time() - t0 < secs:
                    for _ in range(10):
                        assert L.pk_trigger_launch(kinds[kname], 2048, 4000, None, side.cuda_stream) == 0
                        assert L.pk_victim_launch(form, 2048, 2000, inp.data_ptr(), mism.data_ptr(), first.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
                        launches += 1
                    torch.cuda.synchronize()
                m = mism.tolist()
                print("trigger %-12s victim %-52s %5d launches: threads with a wrong LOW half %d, HIGH %d" % (kname, name, launches, m[0], m[2]), flush=True)
        return
    if family != "none":
        import race_screen

        def _load():
            with torch.cuda.stream(torch.cuda.Stream(dev)):
                while not stop:
                    race_screen.screen(reps=2, verbose=False, deep={}, family=family, burst=25)
        threading.Thread(target=_load, daemon=True).start()
        time.sleep(5)
    for form, name in FORMS:
        mism = torch.zeros(4, dtype=torch.int64, device=dev)
        first = torch.zeros(1, dtype=torch.int32, device=dev)
        t0, launches = time.time(), 0
        while time.time() - t0 < secs:
            for _ in range(10):
                rc = L.pk_victim_launch(form, 2048, 2000, inp.data_ptr(), mism.data_ptr(), first.data_ptr(), torch.cuda.current_stream().cuda_stream)
                assert rc == 0, rc
                launches += 1
            torch.cuda.current_stream().synchronize()
        m = mism.tolist()
        print("%-78s %5d launches x 2048 x 256 threads x 2000 operations: threads with a wrong LOW half %d, only HIGH %d, HIGH %d" % (name, launches, m[0], m[1], m[2]), flush=True)
    stop.append(1)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
