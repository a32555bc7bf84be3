#!/usr/bin/env python3
"""Geometry the host cost model picks for the convolution shapes of the paper network (no GPU needed: planning is host code).
DFL_CONVP_DEBUG=1 python docs/experiments/convp_geom.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import dfl_amd  # noqa
from dfl_amd import _native as nat
lib = nat.lib()
lib.dfl_set_math_mode(4)
SH = [(192,32,32,3),(96,32,64,3),(96,64,64,3),(48,64,128,3),(48,128,128,3),(24,128,256,3),(24,256,256,3),(12,256,512,3),(12,512,512,3),(6,512,1024,3),(6,1024,1024,3),
      (12,1024,512,3),(24,512,256,3),(48,256,128,3),(96,128,64,3),(192,64,32,3),(96,32,64,1),(48,64,128,1),(192,64,32,1),(192,32,64,3),(96,64,128,3),(48,128,256,3),(24,256,512,3)]
for (H,ci,co,k) in SH:
    a = nat.ConvArgs()
    a.x = a.w = a.y = 4096
    a.x_bf16 = a.y_bf16 = 1; a.w_split = 2
    a.N, a.Hin, a.Win, a.Cin, a.ldx = 16, H, H, ci, ci
    a.KH = a.KW = k; a.stride = 1; a.pad = k // 2
    a.Hout = a.Wout = H; a.Ntot = a.ldy = co
    a.splits = 0
    sp = lib.dfl_conv_suggest_splits(C.addressof(a))
