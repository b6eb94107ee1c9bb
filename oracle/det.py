"""ctypes front end of oracle/det_ref.c (TEST INFRASTRUCTURE ONLY)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_build", "libdet_ref.so")
_lib = None


def build():
    src = os.path.join(HERE, "det_ref.c")
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", HERE, "_build/libdet_ref.so"])
    return SO


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def sigmoid(x):
    x = f32(x)
    out = np.empty_like(x)
    lib().det_sigmoid_array(_p(x), _p(out), C.c_long(x.size))
    return out


def nade_sample(bias, w_enc, w_dec, tracks, m, D, Hn, temperature, u):
    bias, w_enc, w_dec, u = f32(bias), f32(w_enc), f32(w_dec), f32(u)
    N = bias.shape[0]
    s = np.empty((N, D), np.uint8)
    p = np.empty((N, D), np.float32)
    lib().nade_sample_det(N, D, Hn, tracks, m, _p(bias), bias.shape[1], _p(w_enc), _p(w_dec),
                          C.c_float(-1.0 if temperature is None else temperature), _p(u), _p(s), _p(p))
    return s, p


def rbm_hidden(v, W, bh):
    v, W, bh = f32(v), f32(W), f32(bh)
    N, D = v.shape
    Hn = W.shape[1]
    out = np.empty((N, Hn), np.float32)
    lib().rbm_hidden_det(N, D, Hn, _p(v), _p(W), _p(bh), 0 if bh.shape[0] == 1 else bh.shape[1], _p(out))
    return out


def rbm_visible(h, W, bv):
    h, W, bv = f32(h), f32(W), f32(bv)
    N, Hn = h.shape
    D = W.shape[0]
    out = np.empty((N, D), np.float32)
    lib().rbm_visible_det(N, D, Hn, _p(h), _p(W), _p(bv), 0 if bv.shape[0] == 1 else bv.shape[1], _p(out))
    return out


def rbm_gibbs(v0, W, bh, bv, k, u_h, u_v):
    v0 = np.ascontiguousarray(v0, np.uint8)
    W, bh, bv, u_h, u_v = f32(W), f32(bh), f32(bv), f32(u_h), f32(u_v)
    N, D = v0.shape
    Hn = W.shape[1]
    p_v = np.empty((N, D), np.float32)
    v = np.empty((N, D), np.uint8)
    vb, hb, pb = np.empty((N, D), np.float32), np.empty((N, Hn), np.float32), np.empty((N, Hn), np.float32)
    lib().rbm_gibbs_det(N, D, Hn, k, _p(v0), _p(W), _p(bh), 0 if bh.shape[0] == 1 else bh.shape[1], _p(bv),
                        0 if bv.shape[0] == 1 else bv.shape[1], _p(u_h), _p(u_v), _p(p_v), _p(v), _p(vb), _p(hb), _p(pb))
    return p_v, v
