"""yoloseries_amd — MI355X (gfx950) native hot path of yl-jiang/YOLOSeries.

Layout:  csrc/ (HIP kernels + C ABI, include/yolohip.h)  ·  _lib.py / hipk.py (ctypes binding)
         models/ loss/ trainer/ utils/ (host-side mirror of the reference's Python surface)
"""
__version__ = "0.1.0"
