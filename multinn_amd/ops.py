"""Torch-tensor front end of the C ABI (include/multinn_hip.h).

Every function validates shapes/dtypes/devices on the host (a mis-sized operand must never reach
a kernel), extracts raw device pointers and the current HIP stream, and calls the library.
There is no CPU path: tensors must be on a ROCm device.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import F32, BF16, U8, F16, GEMM_ACCUMULATE, GEMM_A_KBLOCK32


def call(name, *args):
    return _lib.call(name, *args)

_DT = {torch.float32: F32, torch.bfloat16: BF16, torch.uint8: U8, torch.float16: F16}
H16 = (torch.bfloat16, torch.float16)          # the two 16-bit operand flavours (precision "bf16" / "fp16")
DENSITY_SLOTS = 256                             # MNN_DENSITY_SLOTS: u32 partial counts of mnn_density_gate / the piano-roll pass


def dtype_code(t):
    return _DT[t.dtype if isinstance(t, torch.Tensor) else t]


def _stream():
    if not torch.cuda.is_available():
        raise _lib.MnnError("multinn_amd has no CPU path: no ROCm device is available")
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.MnnError("multinn_amd has no CPU path: tensor is on %s" % t.device)
    return C.c_void_p(t.data_ptr())


def _req(cond, msg):
    if not cond:
        raise ValueError(msg)


def _rowmajor(t, what):
    _req(t.dim() == 2 and t.stride(1) == 1, f"{what}: need a 2-D tensor with unit inner stride, got {tuple(t.shape)} {t.stride()}")


def round_up(x, m):
    return (x + m - 1) // m * m


# ------------------------------------------------------------------------------------------------
def gemm_tn(A, B, C_out, bias=None, accumulate=False, split_k=1, a_kblock=False, m_rows=None, k_rows=None):
    """C[M,N] (+)= A[M,K] . B[N,K]^T (+bias).  A,B same dtype (f32/bf16/f16), K-contiguous views.  a_kblock: A is
    given K-blocked, a contiguous [K/32, rows >= M, 32] tensor (element (m, k) at [k // 32, m, k % 32]), 16-bit, K % 64 == 0.
    m_rows / k_rows (int32 device words, compacted ragged batches): how many rows of A / how much of K carry data -- a hint the large-tile
    kernels use to skip the padding (mnn_gemm_tn_rows); rows of C past m_rows are then left unwritten."""
    _rowmajor(B, "gemm B"); _rowmajor(C_out, "gemm C")
    _req(A.dtype == B.dtype and A.dtype in (torch.float32,) + H16, "gemm: A/B must both be f32, bf16 or f16")
    if a_kblock:
        _req(A.dim() == 3 and A.is_contiguous() and A.shape[2] == 32 and A.dtype in H16, "gemm: a K-blocked A is a contiguous 16-bit [K/32, rows, 32]")
        K, M, lda = A.shape[0] * 32, C_out.shape[0], A.shape[1]
        _req(M <= lda and K % 64 == 0, "gemm: K-blocked A: rows >= M, K % 64 == 0")
    else:
        _rowmajor(A, "gemm A")
        M, K = A.shape
        lda = A.stride(0)
    N, K2 = B.shape
    _req(K == K2 and C_out.shape == (M, N), f"gemm: shape mismatch A{tuple(A.shape)} B{tuple(B.shape)} C{tuple(C_out.shape)}")
    _req(C_out.dtype in (torch.float32,) + H16, "gemm: C must be f32/bf16/f16")
    if bias is not None:
        _req(bias.dtype == torch.float32 and bias.numel() == N and bias.is_contiguous(), "gemm: bias must be f32[N]")
    flags = (GEMM_ACCUMULATE if accumulate else 0) | (GEMM_A_KBLOCK32 if a_kblock else 0)
    if m_rows is not None or k_rows is not None:
        for t in (m_rows, k_rows):
            _req(t is None or (t.dtype == torch.int32 and t.numel() >= 1 and t.is_cuda), "gemm: m_rows / k_rows are int32 device words")
        call("mnn_gemm_tn_rows", _stream(), dtype_code(A), M, N, K, _ptr(A), lda, _ptr(B), B.stride(0), _ptr(C_out), C_out.stride(0),
             dtype_code(C_out), _ptr(bias), flags, split_k, _ptr(m_rows), _ptr(k_rows))
        return C_out
    call("mnn_gemm_tn", _stream(), dtype_code(A), M, N, K, _ptr(A), lda, _ptr(B), B.stride(0), _ptr(C_out), C_out.stride(0),
         dtype_code(C_out), _ptr(bias), flags, split_k)
    return C_out


def transpose(src, out):
    """out[C,R] = src[R,C]^T with conversion."""
    _rowmajor(src, "transpose src"); _rowmajor(out, "transpose out")
    R, Cc = src.shape
    _req(out.shape[0] == Cc and out.shape[1] >= R, "transpose: out must be [C, >=R]")
    call("mnn_transpose", _stream(), _ptr(src), dtype_code(src), R, Cc, src.stride(0), _ptr(out), dtype_code(out), out.stride(0))
    return out


def convert2d(src, dst):
    _rowmajor(src, "convert src"); _rowmajor(dst, "convert dst")
    _req(src.shape == dst.shape, "convert2d: shape mismatch")
    call("mnn_convert2d", _stream(), _ptr(src), dtype_code(src), src.stride(0), _ptr(dst), dtype_code(dst), dst.stride(0), src.shape[0], src.shape[1])
    return dst


def ragged_index(lengths, B, T, n_total_dev=None, scale_rows=0.0):
    """Compaction index of a ragged window, built on the device (mnn_ragged_index): returns (idx, inv, hdr) -- idx[k] = time-major row at compact
    position k (valid rows first, time-major order kept), inv = its inverse, hdr int32 [4] = [n_valid | 1 / n_total | loss scale | 1 / loss scale]
    (words 1..3 are f32 bits: hdr.view(torch.float32)).  n_total_dev: f32 [1] valid rows of ALL ranks (None: this rank's)."""
    _req(lengths.dtype == torch.int32 and lengths.numel() == B and lengths.is_cuda and lengths.is_contiguous(), "ragged_index: lengths int32 [B] on the device")
    _req(n_total_dev is None or (n_total_dev.dtype == torch.float32 and n_total_dev.numel() == 1), "ragged_index: n_total_dev f32 [1]")
    dev = lengths.device
    idx = torch.empty(B * T, device=dev, dtype=torch.int32)
    inv = torch.empty(B * T, device=dev, dtype=torch.int32)
    hdr = torch.empty(4, device=dev, dtype=torch.int32)
    call("mnn_ragged_index", _stream(), _ptr(lengths), int(B), int(T), _ptr(n_total_dev), float(scale_rows), _ptr(idx), _ptr(inv), _ptr(hdr))
    return idx, inv, hdr


def rows_gather16(src, idx, hdr, dst, dst_t=None):
    """dst[k] = src[idx[k]] for k < hdr[0], zeros behind (16-bit [N, C] rows); dst_t [C, >= N] (optional): the transposed copy."""
    N, Cc = src.shape
    _req(src.dtype in H16 and dst.dtype == src.dtype and dst.shape == src.shape and src.stride(1) == 1 and dst.stride(1) == 1, "rows_gather16: 16-bit [N, C]")
    _req(dst_t is None or (dst_t.dtype == src.dtype and dst_t.shape[0] == Cc and dst_t.shape[1] >= N and dst_t.stride(1) == 1), "rows_gather16: dst_t [C, >= N]")
    call("mnn_rows_gather16", _stream(), _ptr(src), src.stride(0), _ptr(idx), _ptr(hdr), N, Cc, _ptr(dst), dst.stride(0), _ptr(dst_t),
         dst_t.stride(0) if dst_t is not None else 0)
    return dst


def rows_scatter_f32(src, inv, hdr, dst):
    """dst[row] = src[inv[row]] where inv[row] < hdr[0], zeros elsewhere (f32 [N, C], both contiguous)."""
    _req(src.dtype == torch.float32 and dst.dtype == torch.float32 and src.shape == dst.shape and src.is_contiguous() and dst.is_contiguous(),
         "rows_scatter_f32: contiguous f32 [N, C]")
    call("mnn_rows_scatter_f32", _stream(), _ptr(src), _ptr(inv), _ptr(hdr), src.shape[0], src.shape[1], _ptr(dst))
    return dst


def pianoroll_shift_timemajor(x, lengths, inputs, targets, row_weight, n_valid_total=0, inputs_t=None, count=None, compact=None):
    """x u8 [B,T,D] -> inputs [T,B,ld] (shifted, zero first step), targets u8 [T,B,D], row_weight f32 [T*B];
    inputs_t (bf16 [ld, >= T*B], optional): the transposed copy of inputs, written by the same pass.
    compact = (inv, hdr) of ragged_index (tiled 16-bit pass only): targets and row_weight are written in COMPACT row order, the weights from the
    device-side header (no host-side row count)."""
    _req(x.dtype == torch.uint8 and x.dim() == 3 and x.is_contiguous(), "pianoroll: x must be contiguous u8 [B,T,D]")
    B, T, D = x.shape
    _req(inputs.dim() == 3 and inputs.shape[0] == T and inputs.shape[1] == B and inputs.shape[2] >= D and inputs.is_contiguous(),
         "pianoroll: inputs must be contiguous [T,B,>=D]")
    if targets is not None:
        _req(targets.dtype == torch.uint8 and targets.shape == (T, B, D) and targets.is_contiguous(), "pianoroll: targets u8 [T,B,D]")
    if row_weight is not None:
        _req(row_weight.dtype == torch.float32 and row_weight.numel() == T * B, "pianoroll: row_weight f32 [T*B]")
    _req(compact is None or inputs_t is not None, "pianoroll: compact row order needs the tiled 16-bit pass (inputs_t)")
    if lengths is not None:
        _req(lengths.dtype == torch.int32 and lengths.numel() == B, "pianoroll: lengths int32 [B]")
        _req(n_valid_total > 0 or compact is not None, "pianoroll: n_valid_total required with lengths")
    if inputs_t is not None:
        _req(inputs.dtype in H16 and inputs_t.dtype == inputs.dtype and inputs_t.dim() == 2 and inputs_t.stride(1) == 1
             and inputs_t.shape[0] == inputs.shape[2] and inputs_t.shape[1] >= T * B, "pianoroll: inputs_t must be 16-bit [ld, >=T*B] like inputs")
        call("mnn_pianoroll_shift_timemajor_t", _stream(), _ptr(x), B, T, D, _ptr(lengths), _ptr(inputs), inputs.shape[2], _ptr(inputs_t),
             inputs_t.stride(0), _ptr(targets), _ptr(row_weight), int(n_valid_total), dtype_code(inputs), _ptr(count),
             _ptr(compact[0]) if compact is not None else None, _ptr(compact[1]) if compact is not None else None)
        return count is not None                    # True: `count` now holds the number of set target cells
    call("mnn_pianoroll_shift_timemajor", _stream(), _ptr(x), B, T, D, _ptr(lengths), _ptr(inputs), dtype_code(inputs), inputs.shape[2],
         _ptr(targets), _ptr(row_weight), int(n_valid_total))
    return False


def pianoroll_split_tracks(x, out):
    _req(x.dtype == torch.uint8 and x.dim() == 4 and x.is_contiguous(), "split_tracks: x must be contiguous u8 [B,T,P,M]")
    B, T, P, M = x.shape
    _req(out.dtype == torch.uint8 and out.shape == (M, T, B, P) and out.is_contiguous(), "split_tracks: out u8 [M,T,B,P]")
    call("mnn_pianoroll_split_tracks", _stream(), _ptr(x), B, T, P, M, _ptr(out))
    return out


# ------------------------------------------------------------------------------------------------
def lstm_pack_weights(W, b, n_in, units, wx_t, wh_t, wh_p, wx_p, bias_p):
    _req(W.dtype == torch.float32 and W.shape == (n_in + units, 4 * units) and W.is_contiguous(), "pack: W f32 [(in+u),4u]")
    _req(b.dtype == torch.float32 and b.numel() == 4 * units, "pack: b f32 [4u]")
    ld_in = wx_t.shape[1]
    _req(wx_t.shape == (4 * units, ld_in) and ld_in >= n_in and wx_t.is_contiguous(), "pack: wx_t [4u, ld_in]")
    _req(wh_t.shape == (4 * units, units) and wh_t.is_contiguous(), "pack: wh_t [4u,u]")
    _req(wh_p.shape == (units, 4 * units) and wh_p.is_contiguous(), "pack: wh_p [u,4u]")
    if wx_p is not None:
        _req(wx_p.shape == (n_in, 4 * units) and wx_p.is_contiguous() and wx_p.dtype == wx_t.dtype, "pack: wx_p [in,4u]")
    _req(wx_t.dtype == wh_t.dtype == wh_p.dtype, "pack: dtype mismatch")
    _req(bias_p.dtype == torch.float32 and bias_p.numel() == 4 * units, "pack: bias_p f32 [4u]")
    call("mnn_lstm_pack_weights", _stream(), _ptr(W), _ptr(b), n_in, units, dtype_code(wx_t), ld_in, _ptr(wx_t), _ptr(wh_t), _ptr(wh_p),
         _ptr(wx_p), _ptr(bias_p))


def lstm_rows_gate_minor(wx_t, bias_p, wx_gm, bias_gm):
    """Gate-minor (row unit*4+g) copy of a packed wx_t [4u, ld] and its bias: the layout the persistent recurrence reads xproj in."""
    N4, ld = wx_t.shape
    _req(N4 % 128 == 0 and wx_t.is_contiguous() and wx_gm.shape == wx_t.shape and wx_gm.dtype == wx_t.dtype and wx_gm.is_contiguous(),
         "rows_gate_minor: wx_t / wx_gm [4u, ld]")
    _req(bias_p.dtype == torch.float32 and bias_gm.dtype == torch.float32 and bias_p.numel() == N4 and bias_gm.numel() == N4, "rows_gate_minor: bias f32 [4u]")
    call("mnn_lstm_rows_gate_minor", _stream(), dtype_code(wx_t), N4 // 4, ld, _ptr(wx_t), _ptr(bias_p), _ptr(wx_gm), _ptr(bias_gm))


def lstm_unpack_grads(dwx_t, dwh_t, db_p, n_in, units, dW, db, consume=False):
    """consume=True: the packed sources are zeroed as they are read (persistent accumulators, see mnn_lstm_unpack_grads_consume)."""
    ld_in = dwx_t.shape[1]
    _req(dwx_t.dtype == torch.float32 and dwx_t.shape == (4 * units, ld_in) and dwx_t.is_contiguous(), "unpack: dwx_t f32 [4u,ld]")
    _req(dwh_t.dtype == torch.float32 and dwh_t.shape == (4 * units, units) and dwh_t.is_contiguous(), "unpack: dwh_t f32 [4u,u]")
    _req(dW.dtype == torch.float32 and dW.shape == (n_in + units, 4 * units) and dW.is_contiguous(), "unpack: dW f32 [(in+u),4u]")
    _req(db.numel() == 4 * units and db_p.numel() == 4 * units, "unpack: bias sizes")
    call("mnn_lstm_unpack_grads_consume" if consume else "mnn_lstm_unpack_grads", _stream(), _ptr(dwx_t), _ptr(dwh_t), _ptr(db_p), n_in, units, ld_in,
         _ptr(dW), _ptr(db))


def lstm_unpack_grads_cat(dw_cat, db_p, n_in, units, ld_in, dW, db):
    """dw_cat f32 [4u, ld_in + u] = [dWx^T | dWh^T] (one GEMM over the concatenated operand) -> dW [(in+u), 4u], db [4u], accumulating; the
    packed sources are zeroed as they are read."""
    _req(dw_cat.dtype == torch.float32 and dw_cat.shape == (4 * units, ld_in + units) and dw_cat.is_contiguous(), "unpack_cat: dw_cat f32 [4u, ld+u]")
    _req(dW.dtype == torch.float32 and dW.shape == (n_in + units, 4 * units) and dW.is_contiguous(), "unpack_cat: dW f32 [(in+u),4u]")
    _req(db.numel() == 4 * units and db_p.numel() == 4 * units and db_p.dtype == torch.float32, "unpack_cat: bias sizes")
    call("mnn_lstm_unpack_grads_cat", _stream(), _ptr(dw_cat), _ptr(db_p), n_in, units, ld_in, _ptr(dW), _ptr(db))


def lstm_fused_outputs(dtype, units):
    return bool(_lib.load().mnn_lstm_fused_outputs(dtype_code(dtype), int(units)))


def lstm_seq_fwd(xproj, wh_t, h0, c0, gates, c, h, t_begin=0, t_end=None, hT=None):
    T, B, N4 = xproj.shape
    units = N4 // 4
    _req(xproj.dtype == torch.float32 and xproj.is_contiguous() and units % 32 == 0, "lstm_fwd: xproj f32 [T,B,4u], u%32==0")
    _req(wh_t.shape == (N4, units) and wh_t.is_contiguous() and wh_t.dtype == h.dtype, "lstm_fwd: wh_t [4u,u] in the compute dtype")
    _req(c.dtype == torch.float32 and c.shape == (T, B, units) and c.is_contiguous(), "lstm_fwd: c f32 [T,B,u]")
    _req(h.shape == (T, B, units) and h.is_contiguous(), "lstm_fwd: h [T,B,u]")
    if gates is not None:
        _req(gates.dtype == torch.float32 and gates.shape == (T, B, N4) and gates.is_contiguous(), "lstm_fwd: gates f32 [T,B,4u]")
    if h0 is not None:
        _req(h0.shape == (B, units) and h0.dtype == h.dtype and h0.is_contiguous(), "lstm_fwd: h0")
    if c0 is not None:
        _req(c0.shape == (B, units) and c0.dtype == torch.float32 and c0.is_contiguous(), "lstm_fwd: c0")
    t_end = T if t_end is None else t_end
    _req(0 <= t_begin < t_end <= T, "lstm_fwd: bad step range")
    if hT is not None:
        _req(hT.dim() == 2 and hT.shape[0] == units and hT.stride(1) == 1 and hT.shape[1] >= T * B and hT.dtype == h.dtype, "lstm_fwd: hT [u, >=T*B]")
    call("mnn_lstm_seq_fwd", _stream(), dtype_code(h), T, B, units, int(t_begin), int(t_end), _ptr(xproj), _ptr(wh_t), _ptr(h0), _ptr(c0),
         _ptr(gates), _ptr(c), _ptr(h), _ptr(hT), hT.stride(0) if hT is not None else 0)


def lstm_seq_bwd_workspace(B, units, device):
    return torch.empty(_lib.load().mnn_lstm_seq_bwd_workspace_bytes(B, units), dtype=torch.uint8, device=device)


def lstm_seq_bwd(dh_ext, wh_p, gates, c, c0, dz, dz_T, dh0=None, dc0=None, t_begin=0, t_end=None, ws=None, dzT_t=None, db_p=None):
    T, B, units = dh_ext.shape
    N4 = 4 * units
    _req(dh_ext.dtype == torch.float32 and dh_ext.is_contiguous(), "lstm_bwd: dh_ext f32 [T,B,u]")
    _req(wh_p.shape == (units, N4) and wh_p.is_contiguous(), "lstm_bwd: wh_p [u,4u]")
    _req(gates.shape == (T, B, N4) and gates.dtype == torch.float32 and gates.is_contiguous(), "lstm_bwd: gates")
    _req(c.shape == (T, B, units) and c.dtype == torch.float32 and c.is_contiguous(), "lstm_bwd: c")
    if dz is not None:
        _req(dz.shape == (T, B, N4) and dz.dtype == torch.float32 and dz.is_contiguous(), "lstm_bwd: dz f32 [T,B,4u]")
    if dz_T is not None:
        _req(dz_T.shape == (T, B, N4) and dz_T.dtype == wh_p.dtype and dz_T.is_contiguous(), "lstm_bwd: dz_T")
    t_end = T if t_end is None else t_end
    _req(0 <= t_begin < t_end <= T, "lstm_bwd: bad step range")
    if ws is None:
        _req(t_begin == 0 and t_end == T, "lstm_bwd: chunked calls must share a workspace (lstm_seq_bwd_workspace)")
        ws = lstm_seq_bwd_workspace(B, units, dh_ext.device)
    if dzT_t is not None:
        _req(dzT_t.dim() == 2 and dzT_t.shape[0] == N4 and dzT_t.stride(1) == 1 and dzT_t.shape[1] >= T * B and dzT_t.dtype == wh_p.dtype,
             "lstm_bwd: dzT_t [4u, >=T*B]")
    if db_p is not None:
        _req(db_p.dtype == torch.float32 and db_p.numel() == N4 and db_p.is_contiguous(), "lstm_bwd: db_p f32 [4u]")
    call("mnn_lstm_seq_bwd", _stream(), dtype_code(wh_p), T, B, units, int(t_begin), int(t_end), _ptr(dh_ext), _ptr(wh_p), _ptr(gates),
         _ptr(c), _ptr(c0), _ptr(dz), _ptr(dz_T), _ptr(dh0), _ptr(dc0), _ptr(ws), _ptr(dzT_t), dzT_t.stride(0) if dzT_t is not None else 0,
         _ptr(db_p))


def _step_ok(step_dev):
    _req(step_dev is None or (step_dev.dtype == torch.int32 and step_dev.numel() == 1), "step_dev must be a device int32 scalar")


def _p0(t):
    return t.data_ptr() if t is not None else None


def lstm2_fwd_layer(xproj, wh_t, h0, c0, gates, c, h, hT, y=None, mask=None, wx_t=None, bias_p=None, yT=None, gates_dtype=torch.float32,
                    xproj_dtype=torch.float32):
    """Descriptor of one layer for lstm2_seq_fwd (same tensors as lstm_seq_fwd; bf16, contiguous, time-major).
    y/mask: dropped output and u8 keep mask (keep_prob < 1); wx_t/bias_p: this layer's input projection (layer 2)."""
    _req(h.dim() == 3 and h.dtype in H16 and h.is_contiguous(), "lstm2: h bf16 / f16 [T,B,u]")
    dt = h.dtype                                    # the layer's 16-bit flavour: every 16-bit tensor of the layer has it
    T, B, u = h.shape
    N4 = 4 * u
    _req(xproj is not None or wx_t is not None, "lstm2: xproj may be omitted only for a layer with its own input projection (persistent form)")
    _req(xproj is None or (xproj.dtype == xproj_dtype and xproj.is_contiguous() and xproj.shape == (T, B, N4)), "lstm2: xproj [T,B,4u] of the stated dtype")
    _req(xproj_dtype in (torch.float32, dt), "lstm2: xproj f32 (or the layer's 16-bit type: row-parallel form only)")
    _req(gates_dtype in (torch.float32, dt), "lstm2: gates f32 (or the layer's 16-bit type: row-parallel form)")
    _req(wh_t.shape == (N4, u) and wh_t.is_contiguous() and wh_t.dtype == dt, "lstm2: wh_t [4u,u] in the layer's 16-bit type")
    _req(c.dtype == torch.float32 and c.shape == (T, B, u) and c.is_contiguous(), "lstm2: c")
    _req(gates is None or (gates.dtype == gates_dtype and gates.shape == (T, B, N4) and gates.is_contiguous()), "lstm2: gates")
    _req(h0 is None or (h0.shape == (B, u) and h0.dtype == dt and h0.is_contiguous()), "lstm2: h0")
    _req(c0 is None or (c0.shape == (B, u) and c0.dtype == torch.float32 and c0.is_contiguous()), "lstm2: c0")
    _req(hT is None or (hT.dim() == 2 and hT.shape[0] == u and hT.stride(1) == 1 and hT.shape[1] >= T * B and hT.dtype == dt), "lstm2: hT")
    _req((y is None) == (mask is None), "lstm2: y and mask come together")
    if mask is not None:
        _req(mask.dtype == torch.uint8 and mask.shape == (T, B, u) and mask.is_contiguous(), "lstm2: mask u8 [T,B,u]")
        _req(y.dtype == dt and y.shape == (T, B, u) and y.is_contiguous(), "lstm2: y [T,B,u] in the layer's 16-bit type")
    ld_w = 0
    if wx_t is not None:
        _req(wx_t.dim() == 2 and wx_t.shape[0] == N4 and wx_t.stride(1) == 1 and wx_t.dtype == dt, "lstm2: wx_t [4u, ld] in the layer's 16-bit type")
        _req(bias_p is not None and bias_p.dtype == torch.float32 and bias_p.numel() == N4, "lstm2: bias_p f32 [4u]")
        ld_w = wx_t.stride(0)
    for t in (wh_t, c, h):
        _ptr(t)
    _req(yT is None or (yT.dim() == 2 and yT.shape[0] == u and yT.stride(1) == 1 and yT.shape[1] >= T * B and yT.dtype == dt), "lstm2: yT")
    return _lib.LstmFwdLayer(u, _p0(xproj), _p0(wh_t), _p0(h0), _p0(c0), _p0(gates), _p0(c), _p0(h), _p0(hT), hT.stride(0) if hT is not None else 0,
                             _p0(y), _p0(mask), _p0(wx_t), ld_w, _p0(bias_p), _p0(yT), yT.stride(0) if yT is not None else 0,
                             1 if xproj_dtype != torch.float32 else 0, 1 if dt == torch.float16 else 0)


def lstm2_seq_fwd(T, B, L1, L2, keep_prob, s_begin=0, s_end=None):
    s_end = T + 2 if s_end is None else s_end
    _req(0 <= s_begin < s_end <= T + 2, "lstm2_fwd: bad launch range")
    call("mnn_lstm2_seq_fwd", _stream(), T, B, C.byref(L1), C.byref(L2), float(keep_prob), int(s_begin), int(s_end))


def lstm2_bwd_layer(dh_ext, wh_p, gates, c, c0, dz_T, ws, dzT_t, db_p, mask=None, wx_p=None, gates_dtype=torch.float32):
    T, B, u = c.shape
    N4 = 4 * u
    _req(dh_ext is None or (dh_ext.dtype == torch.float32 and dh_ext.is_contiguous() and dh_ext.shape == (T, B, u)), "lstm2 bwd: dh_ext f32 [T,B,u]")
    _req(wh_p.shape == (u, N4) and wh_p.is_contiguous() and wh_p.dtype in H16, "lstm2 bwd: wh_p bf16 / f16 [u,4u]")
    dt = wh_p.dtype
    _req(gates_dtype in (torch.float32, dt), "lstm2 bwd: gates f32 (or the layer's 16-bit type: row-parallel form)")
    _req(gates.shape == (T, B, N4) and gates.dtype == gates_dtype and gates.is_contiguous(), "lstm2 bwd: gates")
    _req(c.shape == (T, B, u) and c.dtype == torch.float32 and c.is_contiguous(), "lstm2 bwd: c")
    _req(dz_T is None or (dz_T.shape == (T, B, N4) and dz_T.dtype == dt and dz_T.is_contiguous()), "lstm2 bwd: dz_T [T,B,4u] in the layer's 16-bit type")
    kblock = dzT_t is not None and dzT_t.dim() == 3         # [T*B/32, 4u, 32]: the K-blocked layout (row-parallel form only; ld_t = 0 in the ABI)
    _req(dzT_t is None or (kblock and dzT_t.shape == (T * B // 32, N4, 32) and dzT_t.is_contiguous() and dzT_t.dtype == dt and B % 32 == 0)
         or (dzT_t.dim() == 2 and dzT_t.shape[0] == N4 and dzT_t.stride(1) == 1 and dzT_t.shape[1] >= T * B and dzT_t.dtype == dt), "lstm2 bwd: dzT_t")
    _req(db_p is None or (db_p.dtype == torch.float32 and db_p.numel() == N4 and (dzT_t is not None or dz_T is not None)), "lstm2 bwd: db_p")
    _req(ws.numel() >= B * u * 4, "lstm2 bwd: workspace too small")
    _req(mask is None or (mask.dtype == torch.uint8 and mask.shape == (T, B, u) and mask.is_contiguous()), "lstm2 bwd: mask u8 [T,B,u]")
    _req(wx_p is None or (wx_p.dim() == 2 and wx_p.shape[1] == N4 and wx_p.is_contiguous() and wx_p.dtype == dt), "lstm2 bwd: wx_p [n_in,4u]")
    for t in (wh_p, gates, c, ws):
        _ptr(t)
    return _lib.LstmBwdLayer(u, _p0(dh_ext), _p0(wh_p), _p0(gates), _p0(c), _p0(c0), None, _p0(dz_T), _p0(ws), _p0(dzT_t),
                             dzT_t.stride(0) if (dzT_t is not None and not kblock) else 0, _p0(db_p), _p0(mask), _p0(wx_p), 1 if dt == torch.float16 else 0)


def lstm2_seq_bwd(T, B, L1, L2, keep_prob, k_begin=0, k_end=None):
    k_end = T + 2 if k_end is None else k_end
    _req(0 <= k_begin < k_end <= T + 2, "lstm2_bwd: bad launch range")
    call("mnn_lstm2_seq_bwd", _stream(), T, B, C.byref(L1), C.byref(L2), float(keep_prob), int(k_begin), int(k_end))


def lstm2_persist_ok(B, u1, u2):
    """True when the persistent (one launch for all T steps) recurrence can run this shape on this device."""
    return bool(_lib.load().mnn_lstm2_persist_ok(int(B), int(u1), int(u2)))


def lstm2_persist_workspace(T, B, u1, u2, device):
    """Progress flags + exchange area of the persistent recurrence (zeroed here, once)."""
    n = _lib.load().mnn_lstm2_persist_workspace_bytes(int(T), int(B), int(u1), int(u2))
    return torch.zeros(n, dtype=torch.uint8, device=device)


def _ws_ok(ws, T, B, L1, L2):
    _req(ws.dtype == torch.uint8 and ws.is_contiguous() and ws.data_ptr() % 256 == 0
         and ws.numel() >= _lib.load().mnn_lstm2_persist_workspace_bytes(int(T), int(B), L1.units, L2.units),
         "lstm2 persist: workspace must be the tensor of lstm2_persist_workspace(T, B, u1, u2)")


def lstm2_persist_fwd(T, B, L1, L2, keep_prob, ws):
    _ws_ok(ws, T, B, L1, L2)
    call("mnn_lstm2_persist_fwd", _stream(), T, B, C.byref(L1), C.byref(L2), float(keep_prob), _ptr(ws))


def lstm2_persist_bwd(T, B, L1, L2, keep_prob, ws):
    _ws_ok(ws, T, B, L1, L2)
    call("mnn_lstm2_persist_bwd", _stream(), T, B, C.byref(L1), C.byref(L2), float(keep_prob), _ptr(ws))


def lstm2_persist_check(ws, B, u1, u2):
    """Raise if any persistent launch that used this workspace gave up on a bounded spin (synchronises)."""
    st = C.c_int(0)
    call("mnn_lstm2_persist_status", _ptr(ws), int(B), int(u1), int(u2), C.byref(st))
    if st.value != 0:
        raise _lib.MnnError("persistent LSTM launch timed out waiting for a neighbouring workgroup (grid not co-resident?)")


def lstm_rowpar_ok(B, units):
    """True when the row-parallel persistent recurrence (one layer per launch, weights in LDS, a wave per row tile) covers this shape."""
    return bool(_lib.load().mnn_lstm_rowpar_ok(int(B), int(units)))


def lstm_rowpar_workspace(T, B, units, device):
    """Progress flags + exchange area of one layer's row-parallel launches (zeroed here, once; forward and backward share it)."""
    return torch.zeros(_lib.load().mnn_lstm_rowpar_workspace_bytes(int(T), int(B), int(units)), dtype=torch.uint8, device=device)


def _rp_ws_ok(ws, T, B, units):
    _req(ws.dtype == torch.uint8 and ws.is_contiguous() and ws.data_ptr() % 256 == 0
         and ws.numel() >= _lib.load().mnn_lstm_rowpar_workspace_bytes(int(T), int(B), int(units)),
         "lstm rowpar: workspace must be the tensor of lstm_rowpar_workspace(T, B, units)")


def lstm_rowpar_fwd(T, B, L, keep_prob, ws):
    """L: descriptor of lstm2_fwd_layer (xproj gate-minor incl. bias; no h0 / c0)."""
    _rp_ws_ok(ws, T, B, L.units)
    call("mnn_lstm_rowpar_fwd", _stream(), T, B, C.byref(L), float(keep_prob), _ptr(ws))


def lstm_rowpar_bwd(T, B, L, keep_prob, ws):
    """L: descriptor of lstm2_bwd_layer (dh_ext required; dz_T = optional row-major bf16 dz for the input-gradient GEMM)."""
    _rp_ws_ok(ws, T, B, L.units)
    call("mnn_lstm_rowpar_bwd", _stream(), T, B, C.byref(L), float(keep_prob), _ptr(ws))


def lstm_resident_ok(B, units):
    """True when the CU-resident recurrence (a 256-unit layer's whole recurrent matrix on every CU, four batch rows per workgroup) covers this shape."""
    return bool(_lib.load().mnn_lstm_resident_ok(int(B), int(units)))


def lstm_resident_fwd(T, B, L, keep_prob):
    """L: descriptor of lstm2_fwd_layer, as for lstm_rowpar_fwd with a 16-bit xproj; no workspace."""
    call("mnn_lstm_resident_fwd", _stream(), T, B, C.byref(L), float(keep_prob))


def lstm_resident_bwd(T, B, L, keep_prob):
    """L: descriptor of lstm2_bwd_layer, as for lstm_rowpar_bwd; no workspace."""
    call("mnn_lstm_resident_bwd", _stream(), T, B, C.byref(L), float(keep_prob))


def lstm_cluster_ok(B, units):
    """True when the cluster form of the CU-resident recurrence (512 units: eight CUs share 32 rows, all weights in registers) covers this shape."""
    return bool(_lib.load().mnn_lstm_cluster_ok(int(B), int(units)))


def lstm_cluster_bwd_ok(B, units):
    """True when the cluster BACKWARD may run: the shape is covered and every cluster of the launch is dealt onto one XCD (asked on the host by one
    probe launch per device and batch size, cached in the library; False under MNN_PERSIST_NO_LOCAL).  False: take lstm_rowpar_bwd."""
    return bool(_lib.load().mnn_lstm_cluster_bwd_ok(int(B), int(units)))


def lstm_recurrence_multi(kind, T, B, descs, keep_prob, wss=None):
    """One launch for several independent layers of one shape (the per-track generators of the jamming mode).  kind: 'resident_fwd' |
    'resident_bwd' | 'cluster_fwd' | 'cluster_bwd'; descs: the layers' descriptors (lstm2_fwd_layer / lstm2_bwd_layer); wss: cluster forms: one
    lstm_rowpar_workspace per job."""
    n = len(descs)
    _req(1 <= n <= 8, "lstm_recurrence_multi: 1..8 jobs")
    fwd = kind.endswith("fwd")
    arr = ((_lib.LstmFwdLayer if fwd else _lib.LstmBwdLayer) * n)(*descs)
    if kind.startswith("resident"):
        call("mnn_lstm_resident_%s_multi" % ("fwd" if fwd else "bwd"), _stream(), T, B, n, arr, float(keep_prob))
    else:
        _req(wss is not None and len(wss) == n, "lstm_recurrence_multi: one workspace per job")
        for ws, d in zip(wss, descs):
            _rp_ws_ok(ws, T, B, d.units)
        ptrs = (C.c_void_p * n)(*[ws.data_ptr() for ws in wss])
        call("mnn_lstm_cluster_%s_multi" % ("fwd" if fwd else "bwd"), _stream(), T, B, n, arr, float(keep_prob), ptrs)
    return arr            # (keeps the by-value copies alive until the call has returned)


def lstm_cluster_bwd_multi_ok(B, units, njobs):
    return bool(_lib.load().mnn_lstm_cluster_bwd_multi_ok(int(B), int(units), int(njobs)))


def lstm_cluster_fwd(T, B, L, keep_prob, ws):
    """L: descriptor of lstm2_fwd_layer, as for lstm_rowpar_fwd with a 16-bit xproj; ws: the tensor of lstm_rowpar_workspace(T, B, 512)."""
    _rp_ws_ok(ws, T, B, L.units)
    call("mnn_lstm_cluster_fwd", _stream(), T, B, C.byref(L), float(keep_prob), _ptr(ws))


def lstm_cluster_bwd(T, B, L, keep_prob, ws):
    """L: descriptor of lstm2_bwd_layer, as for lstm_rowpar_bwd; ws: the tensor of lstm_rowpar_workspace(T, B, 512); T >= 4."""
    _rp_ws_ok(ws, T, B, L.units)
    call("mnn_lstm_cluster_bwd", _stream(), T, B, C.byref(L), float(keep_prob), _ptr(ws))


def lstm_rowpar_check(ws):
    st = C.c_int(0)
    call("mnn_lstm_rowpar_status", _ptr(ws), C.byref(st))
    if st.value != 0:
        raise _lib.MnnError("row-parallel LSTM launch timed out waiting for a neighbouring wave (grid not co-resident?)")


def dropout_mask(mask, keep_prob, seed, row0, layer, step_dev=None):
    T, B, u = mask.shape
    _req(mask.dtype == torch.uint8 and mask.is_contiguous() and u % 4 == 0 and 0 < keep_prob < 1, "dropout_mask: u8 [T,B,u], 0<kp<1")
    _step_ok(step_dev)
    call("mnn_dropout_mask", _stream(), _ptr(mask), T, B, u, float(keep_prob), int(seed), _ptr(step_dev), int(row0), int(layer))


def dropout_fwd(h, y, keep_prob, seed, row0, layer, step_dev=None, t_offset=0):
    T, B, u = h.shape
    _req(h.is_contiguous() and y.is_contiguous() and h.shape == y.shape and h.dtype == y.dtype and u % 4 == 0, "dropout_fwd: shapes")
    _step_ok(step_dev)
    call("mnn_dropout_fwd", _stream(), dtype_code(h), _ptr(h), _ptr(y), T, B, u, float(keep_prob), int(seed), _ptr(step_dev), int(row0),
         int(layer), int(t_offset))


def dropout_bwd(dy, dh, keep_prob, seed, row0, layer, accumulate=False, step_dev=None, t_offset=0):
    T, B, u = dy.shape
    _req(dy.dtype == torch.float32 and dh.dtype == torch.float32 and dy.is_contiguous() and dh.is_contiguous() and dy.shape == dh.shape,
         "dropout_bwd: f32 [T,B,u]")
    _step_ok(step_dev)
    call("mnn_dropout_bwd", _stream(), _ptr(dy), _ptr(dh), T, B, u, float(keep_prob), int(seed), _ptr(step_dev), int(row0), int(layer),
         int(accumulate), int(t_offset))


# ------------------------------------------------------------------------------------------------
def _nade_check(tracks, N, D, Hn, v, bias, w_enc, w_dec):
    _req(v.dtype == torch.uint8 and v.is_contiguous() and v.numel() == tracks * N * D, "nade: v must be contiguous u8 [tracks,N,D]")
    _req(bias.dtype == torch.float32 and bias.dim() == 2 and bias.shape[0] == N and bias.stride(1) == 1
         and bias.shape[1] >= tracks * (Hn + D), "nade: bias f32 [N, >=tracks*(Hn+D)]")
    for w in (w_enc, w_dec):
        _req(w.dtype == torch.float32 and w.is_contiguous() and w.numel() == tracks * D * Hn, "nade: weights f32 [tracks,D,Hn]")
    _req(0 < Hn <= 256, "nade: Hn must be in 1..256")


def nade_logprob_fwd(v, bias, w_enc, w_dec, tracks, D, Hn, row_weight=None, nll=None, cond_p=None, d_bias=None, a_final=None, n_rows_dev=None,
                     gate=None, run_if=0, unsafe=None):
    N = bias.shape[0]
    _nade_check(tracks, N, D, Hn, v, bias, w_enc, w_dec)
    if nll is not None:
        _req(nll.dtype == torch.float32 and nll.numel() == tracks * N and nll.is_contiguous(), "nade: nll f32 [tracks,N]")
    if cond_p is not None:
        _req(cond_p.dtype == torch.float32 and cond_p.numel() == tracks * N * D and cond_p.is_contiguous(), "nade: cond_p f32 [tracks,N,D]")
    if d_bias is not None:
        _req(row_weight is not None and d_bias.shape == bias.shape and d_bias.stride() == bias.stride() and d_bias.dtype == torch.float32,
             "nade: d_bias must mirror bias and needs row_weight")
    if row_weight is not None:
        _req(row_weight.dtype == torch.float32 and row_weight.numel() == N, "nade: row_weight f32 [N]")
    if a_final is not None:
        _req(a_final.dtype == torch.float32 and a_final.numel() == tracks * N * Hn and a_final.is_contiguous(), "nade: a_final f32 [tracks,N,Hn]")
    _req(unsafe is None or (unsafe.dtype == torch.int32 and unsafe.numel() == 1), "nade: unsafe int32[1]")
    if n_rows_dev is not None or gate is not None or unsafe is not None:   # compacted ragged batch / a density-gated launch: the gated entry point (gate NULL: always runs)
        call("mnn_nade_logprob_fwd_gated", _stream(), tracks, N, D, Hn, _ptr(v), N * D, _ptr(bias), bias.stride(0), _ptr(w_enc), _ptr(w_dec),
             _ptr(row_weight), _ptr(nll), _ptr(cond_p), _ptr(d_bias), _ptr(a_final), _ptr(gate), int(run_if), _ptr(n_rows_dev), _ptr(unsafe))
        return
    call("mnn_nade_logprob_fwd", _stream(), tracks, N, D, Hn, _ptr(v), N * D, _ptr(bias), bias.stride(0), _ptr(w_enc), _ptr(w_dec),
         _ptr(row_weight), _ptr(nll), _ptr(cond_p), _ptr(d_bias), _ptr(a_final))


def nade_mfma_ok(Hn):
    """True when the matrix-core NADE kernels cover this hidden width."""
    return bool(_lib.load().mnn_nade_mfma_ok(int(Hn)))


def nade_logprob_fwd_mfma(v, bias, w_enc, w_dec_bf, tracks, D, Hn, row_weight=None, nll=None, cond_p=None, d_bias=None, a_final=None):
    """nade_logprob_fwd with the decoder dot products on MFMA (bf16 operands): w_dec_bf is the bf16 copy of w_dec."""
    N = bias.shape[0]
    _nade_check(tracks, N, D, Hn, v, bias, w_enc, w_enc)
    _req(w_dec_bf.dtype == torch.bfloat16 and w_dec_bf.is_contiguous() and w_dec_bf.numel() == tracks * D * Hn, "nade mfma: w_dec_bf bf16 [tracks,D,Hn]")
    if nll is not None:
        _req(nll.dtype == torch.float32 and nll.numel() == tracks * N and nll.is_contiguous(), "nade: nll f32 [tracks,N]")
    if cond_p is not None:
        _req(cond_p.dtype == torch.float32 and cond_p.numel() == tracks * N * D and cond_p.is_contiguous(), "nade: cond_p f32 [tracks,N,D]")
    if d_bias is not None:
        _req(row_weight is not None and d_bias.shape == bias.shape and d_bias.stride() == bias.stride() and d_bias.dtype == torch.float32,
             "nade: d_bias must mirror bias and needs row_weight")
    if row_weight is not None:
        _req(row_weight.dtype == torch.float32 and row_weight.numel() == N, "nade: row_weight f32 [N]")
    if a_final is not None:
        _req(a_final.dtype == torch.float32 and a_final.numel() == tracks * N * Hn and a_final.is_contiguous(), "nade: a_final f32 [tracks,N,Hn]")
    call("mnn_nade_logprob_fwd_mfma", _stream(), tracks, N, D, Hn, _ptr(v), N * D, _ptr(bias), bias.stride(0), _ptr(w_enc), _ptr(w_dec_bf),
         _ptr(row_weight), _ptr(nll), _ptr(cond_p), _ptr(d_bias), _ptr(a_final))


def nade_f32_pack(w_dec, out):
    """w_dec f32 [rows, Hn] -> out (an f32 [rows, Hn] buffer holding f16 [rows][hi | lo][Hn], hi = f16(w), lo = f16(w - hi)): the decoder
    operand of the split-operand matrix-core NADE forward (precision "fp16")."""
    _rowmajor(w_dec, "f32_pack w_dec"); _rowmajor(out, "f32_pack out")
    rows, Hn = w_dec.shape
    _req(w_dec.dtype == torch.float32 and w_dec.is_contiguous() and out.dtype == torch.float32 and out.is_contiguous()
         and out.shape == (rows, Hn) and Hn % 8 == 0, "f32_pack: w_dec f32 [rows,Hn] -> out f32 [rows,Hn], Hn % 8 == 0")
    call("mnn_nade_f32_pack", _stream(), _ptr(w_dec), rows, Hn, _ptr(out))
    return out


def nade_logprob_fwd_auto(v, bias, w_enc, w_dec, w_dec_bf, tracks, D, Hn, gate, count, dense_above=0.07, row_weight=None, nll=None,
                          cond_p=None, d_bias=None, a_final=None, exact=False, counted=False, n_rows_dev=None, unsafe=None):
    """16-bit compute modes (exact=True: the split-operand hi + lo form of fp16 mode): the matrix-core form of the scan when at most `dense_above` of the cells of v are active, the f32 vector
    form otherwise; decided on the device (mnn_density_gate), both launches issued (one returns at once).  gate int32[1], count: a zeroed
    int32[1] scratch word (left zero)."""
    N = bias.shape[0]
    _nade_check(tracks, N, D, Hn, v, bias, w_enc, w_dec)
    _req(w_dec_bf.dtype == (torch.float32 if exact else torch.bfloat16) and w_dec_bf.is_contiguous() and w_dec_bf.numel() == tracks * D * Hn,
         "nade auto: w_dec_bf bf16 [tracks,D,Hn] (exact form: the f32-sized [tracks,D,Hn] output of nade_f32_pack)")
    _req(gate is None or (gate.dtype == torch.int32 and gate.numel() == 1 and count.dtype == torch.int32 and count.numel() == DENSITY_SLOTS and count.is_contiguous()),
         "nade auto: gate int32[1], count int32[DENSITY_SLOTS]")
    for t, n in ((nll, tracks * N), (cond_p, tracks * N * D), (a_final, tracks * N * Hn)):
        _req(t is None or (t.dtype == torch.float32 and t.numel() == n and t.is_contiguous()), "nade auto: f32 outputs")
    if d_bias is not None:
        _req(row_weight is not None and d_bias.shape == bias.shape and d_bias.stride() == bias.stride() and d_bias.dtype == torch.float32,
             "nade: d_bias must mirror bias and needs row_weight")
    if row_weight is not None:
        _req(row_weight.dtype == torch.float32 and row_weight.numel() == N, "nade: row_weight f32 [N]")
    common = (tracks, N, D, Hn, _ptr(v), N * D, _ptr(bias), bias.stride(0), _ptr(w_enc))
    tail = (_ptr(row_weight), _ptr(nll), _ptr(cond_p), _ptr(d_bias), _ptr(a_final), _ptr(gate))
    mfma = "mnn_nade_logprob_fwd_mfma_f32" if exact else "mnn_nade_logprob_fwd_mfma_gated"
    nr = _ptr(n_rows_dev)                           # compacted ragged batch: the scans skip workgroups of padding (ragged_index)
    if gate is None:                                # gate off: always the matrix-core form
        call(mfma, _stream(), *common, _ptr(w_dec_bf), *tail, 0, nr)
        return
    # counted: `count` already holds the set cells of v (the piano-roll pass counted them while writing v): only the decision kernel runs
    call("mnn_density_gate", _stream(), None if counted else _ptr(v), v.numel(), int(dense_above * v.numel()), _ptr(gate), _ptr(count))
    call(mfma, _stream(), *common, _ptr(w_dec_bf), *tail, 0, nr)
    _req(unsafe is None or (unsafe.dtype == torch.int32 and unsafe.numel() == 1), "nade auto: unsafe int32[1]")
    call("mnn_nade_logprob_fwd_gated", _stream(), *common, _ptr(w_dec), *tail, 1, nr, _ptr(unsafe))


def nade_logprob_bwd(v, bias, w_enc, w_dec, tracks, D, Hn, a_final, d_bias, d_w_enc, d_w_dec, n_rows_dev=None, unsafe=None):
    """unsafe (int32[1]): the counter the forward's density-gated dense launch left (nade_logprob_fwd_auto / nade_logprob_fwd) -- 0 (the dense
    form ran and no wave passed |a| = 40): the scan advances exp(-a) multiplicatively (no exponential per flip; decided on the device, both
    instantiations launched); None: the direct form."""
    _req(unsafe is None or (unsafe.dtype == torch.int32 and unsafe.numel() == 1), "nade bwd: unsafe int32[1]")
    N = bias.shape[0]
    _nade_check(tracks, N, D, Hn, v, bias, w_enc, w_dec)
    _req(d_bias.shape == bias.shape and d_bias.stride() == bias.stride() and d_bias.dtype == torch.float32, "nade bwd: d_bias")
    for w in (d_w_enc, d_w_dec):
        _req(w.dtype == torch.float32 and w.is_contiguous() and w.numel() == tracks * D * Hn, "nade bwd: grad weights f32 [tracks,D,Hn]")
    _req(a_final.dtype == torch.float32 and a_final.numel() == tracks * N * Hn and a_final.is_contiguous(), "nade bwd: a_final f32 [tracks,N,Hn]")
    call("mnn_nade_logprob_bwd", _stream(), tracks, N, D, Hn, _ptr(v), N * D, _ptr(bias), bias.stride(0), _ptr(w_enc), _ptr(w_dec),
         _ptr(a_final), _ptr(d_bias), _ptr(d_w_enc), _ptr(d_w_dec), _ptr(n_rows_dev), _ptr(unsafe))


def nade_sample(bias, w_enc, w_dec, tracks, D, Hn, temperature, seed, row0, sub, samples, track_minor=False, nll=None):
    """samples u8 [N, tracks*D]; feature index m*D+i (track_minor False) or i*tracks+m (True, rnn_multinade.py:313-314)."""
    N = bias.shape[0]
    _req(bias.dtype == torch.float32 and bias.dim() == 2 and bias.stride(1) == 1 and bias.shape[1] >= tracks * (Hn + D), "sample: bias")
    _req(samples.dtype == torch.uint8 and samples.shape == (N, tracks * D) and samples.is_contiguous(), "sample: samples u8 [N,tracks*D]")
    for w in (w_enc, w_dec):
        _req(w.dtype == torch.float32 and w.is_contiguous() and w.numel() == tracks * D * Hn, "sample: weights")
    if nll is not None:
        _req(nll.dtype == torch.float32 and nll.numel() == tracks * N, "sample: nll")
    ts, es = (1, tracks) if track_minor else (D, 1)
    call("mnn_nade_sample", _stream(), tracks, N, D, Hn, _ptr(bias), bias.stride(0), _ptr(w_enc), _ptr(w_dec),
         float(-1.0 if temperature is None else temperature), int(seed), int(row0), int(sub), _ptr(samples), ts, tracks * D, es, _ptr(nll))


def nade_sample_multi(jobs, D, Hn, temperature, row0, sub):
    """mnn_nade_sample for up to 8 single-NADE generators in ONE launch.  job = dict(bias f32 [N, >= Hn + D] (the generator's Dense output),
    w_enc / w_dec f32 [D, Hn], seed, samples = a u8 [N, D] VIEW (any strides, the same for every job: e.g. out[:, s, :, m] of a
    [B, steps, P, M] piano-roll), nll f32 [N] or None)."""
    _req(1 <= len(jobs) <= 8, "nade_sample_multi: 1..8 jobs")
    arr = (_lib.NadeSampleJob * len(jobs))()
    N = jobs[0]["bias"].shape[0]
    rs, es = jobs[0]["samples"].stride()
    for a, j in zip(arr, jobs):
        b, smp, nll = j["bias"], j["samples"], j.get("nll")
        _req(b.dtype == torch.float32 and b.dim() == 2 and b.stride(1) == 1 and b.shape == (N, b.shape[1]) and b.shape[1] >= Hn + D, "sample_multi: bias")
        _req(smp.dtype == torch.uint8 and tuple(smp.shape) == (N, D) and smp.stride() == (rs, es) and es >= 1, "sample_multi: samples u8 [N, D] views of equal strides")
        for w in (j["w_enc"], j["w_dec"]):
            _req(w.dtype == torch.float32 and w.is_contiguous() and w.numel() == D * Hn, "sample_multi: weights f32 [D, Hn]")
        _req(nll is None or (nll.dtype == torch.float32 and nll.is_contiguous() and nll.numel() == N), "sample_multi: nll f32 [N]")
        a.bias, a.ld_bias, a.w_enc, a.w_dec, a.seed, a.samples, a.nll = _ptr(b), b.stride(0), _ptr(j["w_enc"]), _ptr(j["w_dec"]), int(j["seed"]), _ptr(smp), _ptr(nll)
    call("mnn_nade_sample_multi", _stream(), len(jobs), arr, N, D, Hn, float(-1.0 if temperature is None else temperature), int(row0), int(sub), rs, es)


# ------------------------------------------------------------------------------------------------
def _ldb(b, n):
    _req(b.dtype == torch.float32 and b.dim() == 2 and b.stride(1) == 1 and b.shape[1] >= n, "rbm: bias must be f32 [N or 1, n]")
    return 0 if b.shape[0] == 1 else b.stride(0)


def rbm_workspace(D, Hn, device):
    return torch.empty(_lib.load().mnn_rbm_workspace_bytes(D, Hn), dtype=torch.uint8, device=device)


def rbm_gibbs(v0, W, bh, bv, k, seed, row0=0, row_ids=None, sub0=0, p_v=None, v_out=None, seed_step=None):
    """seed_step (int32 device scalar, optional): added to `seed` on the device (graph-replay safe step-dependent draws)."""
    N, D = v0.shape
    Hn = W.shape[1]
    _req(v0.dtype == torch.uint8 and v0.is_contiguous(), "gibbs: v0 u8 [N,D]")
    _req(W.dtype == torch.float32 and W.shape == (D, Hn) and W.is_contiguous(), "gibbs: W f32 [D,Hn]")
    _req(bh.shape[0] in (1, N) and bv.shape[0] in (1, N), "gibbs: bias rows")
    if p_v is not None:
        _req(p_v.dtype == torch.float32 and p_v.shape == (N, D) and p_v.is_contiguous(), "gibbs: p_v f32 [N,D]")
    if v_out is not None:
        _req(v_out.dtype == torch.uint8 and v_out.shape == (N, D) and v_out.is_contiguous(), "gibbs: v_out u8 [N,D]")
    if row_ids is not None:
        _req(row_ids.dtype == torch.int32 and row_ids.numel() == N, "gibbs: row_ids int32 [N]")
    ws = rbm_workspace(D, Hn, v0.device)
    if seed_step is not None:
        _req(seed_step.dtype == torch.int32 and seed_step.numel() == 1, "gibbs: seed_step int32 [1]")
        call("mnn_rbm_gibbs_stepped", _stream(), N, D, Hn, int(k), _ptr(v0), _ptr(W), _ptr(bh), _ldb(bh, Hn), _ptr(bv), _ldb(bv, D), int(seed),
             int(row0), _ptr(row_ids), int(sub0), _ptr(p_v), _ptr(v_out), _ptr(ws), _ptr(seed_step))
        return
    call("mnn_rbm_gibbs", _stream(), N, D, Hn, int(k), _ptr(v0), _ptr(W), _ptr(bh), _ldb(bh, Hn), _ptr(bv), _ldb(bv, D), int(seed), int(row0),
         _ptr(row_ids), int(sub0), _ptr(p_v), _ptr(v_out), _ptr(ws))


def rbm_hidden(v, W, bh, stream_id, seed, row0, sub, p_h=None, h=None):
    N, D = v.shape
    Hn = W.shape[1]
    _req(v.dtype in (torch.uint8, torch.float32) and v.is_contiguous() and W.shape == (D, Hn) and W.is_contiguous(), "rbm_hidden: shapes")
    _req(bh.shape[0] in (1, N), "rbm_hidden: bias rows")
    for t, dt in ((p_h, torch.float32), (h, torch.uint8)):
        if t is not None:
            _req(t.dtype == dt and t.shape == (N, Hn) and t.is_contiguous(), "rbm_hidden: outputs [N,Hn]")
    call("mnn_rbm_hidden", _stream(), N, D, Hn, _ptr(v), dtype_code(v), _ptr(W), _ptr(bh), _ldb(bh, Hn), int(stream_id), int(seed), int(row0),
         int(sub), _ptr(p_h), _ptr(h))


def rbm_visible(h, W, bv, stream_id, seed, row0, sub, p_v=None, v=None):
    N, Hn = h.shape
    D = W.shape[0]
    _req(h.dtype in (torch.uint8, torch.float32) and h.is_contiguous() and W.shape == (D, Hn) and W.is_contiguous(), "rbm_visible: shapes")
    _req(bv.shape[0] in (1, N), "rbm_visible: bias rows")
    for t, dt in ((p_v, torch.float32), (v, torch.uint8)):
        if t is not None:
            _req(t.dtype == dt and t.shape == (N, D) and t.is_contiguous(), "rbm_visible: outputs [N,D]")
    ws = rbm_workspace(D, Hn, h.device)
    call("mnn_rbm_visible", _stream(), N, D, Hn, _ptr(h), dtype_code(h), _ptr(W), _ptr(bv), _ldb(bv, D), int(stream_id), int(seed), int(row0),
         int(sub), _ptr(p_v), _ptr(v), _ptr(ws))


def rbm_free_energy(v, W, bh, bv, F, p_h=None):
    """p_h (optional, f32 [N, Hn]): also receives sigmoid(v W + bh) -- the hidden activations the free-energy gradient needs."""
    N, D = v.shape
    Hn = W.shape[1]
    _req(p_h is None or (p_h.dtype == torch.float32 and p_h.is_contiguous() and tuple(p_h.shape) == (N, Hn)), "free_energy: p_h f32 [N, Hn]")
    _req(v.dtype == torch.uint8 and v.is_contiguous() and W.shape == (D, Hn) and W.is_contiguous(), "free_energy: shapes")
    _req(F.dtype == torch.float32 and F.numel() == N, "free_energy: F f32 [N]")
    _req(bh.shape[0] in (1, N) and bv.shape[0] in (1, N), "free_energy: bias rows")
    call("mnn_rbm_free_energy", _stream(), N, D, Hn, _ptr(v), _ptr(W), _ptr(bh), _ldb(bh, Hn), _ptr(bv), _ldb(bv, D), _ptr(F), _ptr(p_h))
    return F


def rbm_cd_bias_delta(v, p_v, h, p_h, scale, dbv, dbh):
    """dbv += scale * colsum(v - p_v), dbh += scale * colsum(h - p_h) (rbm.py:318-327); v/h u8, p_* f32, outputs f32 (zeroed by the caller)."""
    N, D = v.shape
    Hn = h.shape[1]
    _req(v.dtype == torch.uint8 and h.dtype == torch.uint8 and v.is_contiguous() and h.is_contiguous() and h.shape[0] == N, "cd_bias_delta: v/h u8 [N,.]")
    _req(p_v.dtype == torch.float32 and p_v.shape == (N, D) and p_v.is_contiguous() and p_h.dtype == torch.float32 and p_h.shape == (N, Hn)
         and p_h.is_contiguous(), "cd_bias_delta: p_v / p_h f32 [N,.]")
    _req(dbv.dtype == torch.float32 and dbv.numel() == D and dbv.is_contiguous() and dbh.dtype == torch.float32 and dbh.numel() == Hn
         and dbh.is_contiguous(), "cd_bias_delta: outputs f32 [D] / [Hn]")
    call("mnn_rbm_cd_bias_delta", _stream(), N, D, Hn, _ptr(v), _ptr(p_v), _ptr(h), _ptr(p_h), float(scale), _ptr(dbv), _ptr(dbh))


def sigmoid_grad(dy, y, dz):
    """dz = dy * y * (1 - y) (f32, same shapes; dz may alias dy)."""
    n = y.numel()
    for t in (dy, y, dz):
        _req(t.dtype == torch.float32 and t.is_contiguous() and t.numel() == n, "sigmoid_grad: contiguous f32 buffers of equal size")
    call("mnn_sigmoid_grad_f32", _stream(), n, _ptr(dy), _ptr(y), _ptr(dz))
    return dz


def rbm_cd_rows(v, v_s, sv, ss, rw, scale, d_out, pos, neg):
    """Rows of the LSTM-RBM cost gradient: d_out[:, :Hn] = w (ss - sv), d_out[:, Hn:Hn+D] = w (v_s - v), padding zeroed; pos = w ss, neg = -w sv
    (w = rw * scale).  v / v_s u8 [N,D], sv / ss f32 [N,Hn], d_out f32 [N, ld >= Hn + D] (rnn_rbm.py:113-126, rbm.py:229)."""
    N, D = v.shape
    Hn = sv.shape[1]
    _req(v.dtype == torch.uint8 and v_s.dtype == torch.uint8 and v.is_contiguous() and v_s.is_contiguous() and v_s.shape == (N, D), "cd_rows: v / v_s u8 [N,D]")
    for t in (sv, ss, pos, neg):
        _req(t.dtype == torch.float32 and t.shape == (N, Hn) and t.is_contiguous(), "cd_rows: sv / ss / pos / neg f32 [N,Hn]")
    _req(rw.dtype == torch.float32 and rw.numel() == N and rw.is_contiguous(), "cd_rows: row weights f32 [N]")
    _req(d_out.dtype == torch.float32 and d_out.dim() == 2 and d_out.shape[0] == N and d_out.is_contiguous() and d_out.shape[1] >= Hn + D, "cd_rows: d_out f32 [N, ld]")
    call("mnn_rbm_cd_rows", _stream(), N, D, Hn, d_out.shape[1], _ptr(v), _ptr(v_s), _ptr(sv), _ptr(ss), _ptr(rw), float(scale), _ptr(d_out), _ptr(pos), _ptr(neg))


def rbm_visible_bias_init(colsum, count, bv):
    """bv[d] = log(1e-6 + p/(1-p)), p = colsum[d]/count (rbm.py:286-297)."""
    D = colsum.numel()
    _req(colsum.dtype == torch.float32 and colsum.is_contiguous() and bv.dtype == torch.float32 and bv.numel() == D and bv.is_contiguous(),
         "visible_bias_init: f32 [D] buffers")
    call("mnn_rbm_visible_bias_init", _stream(), D, _ptr(colsum), float(count), _ptr(bv))


def axpby(a, x, b, y, out):
    """out = a*x + b*y over flat f32 buffers (out may alias x or y; y None with b == 0)."""
    n = x.numel()
    for t in (x, out) + (() if y is None else (y,)):
        _req(t.dtype == torch.float32 and t.is_contiguous() and t.numel() == n, "axpby: contiguous f32 buffers of equal size")
    _req(y is not None or b == 0, "axpby: y missing")
    call("mnn_axpby_f32", _stream(), n, float(a), _ptr(x), float(b), _ptr(y), _ptr(out))
    return out


# ------------------------------------------------------------------------------------------------
def sumsq(x, out):
    _req(x.dtype == torch.float32 and x.is_contiguous() and out.dtype == torch.float32, "sumsq: f32")
    call("mnn_sumsq", _stream(), _ptr(x), x.numel(), _ptr(out))


def weighted_sum(x, w, out):
    _req(x.dtype == torch.float32 and x.is_contiguous() and (w is None or (w.numel() == x.numel() and w.dtype == torch.float32)), "weighted_sum")
    call("mnn_weighted_sum", _stream(), _ptr(x), _ptr(w), x.numel(), _ptr(out))


def clip_adam_step(theta, grad, m, v, sumsq_buf, clip_norm, lr, beta1, beta2, eps, step, sgd=False, step_dev=None, skipped=None):
    n = theta.numel()
    for t in (theta, grad) + (() if sgd else (m, v)):
        _req(t.dtype == torch.float32 and t.is_contiguous() and t.numel() == n, "adam: flat f32 buffers of equal size")
    call("mnn_clip_adam_step", _stream(), _ptr(theta), _ptr(grad), _ptr(m), _ptr(v), n, _ptr(sumsq_buf), float(clip_norm), float(lr),
         float(beta1), float(beta2), float(eps), int(step), _ptr(step_dev), int(sgd), _ptr(skipped))


def step_increment(step_dev, sumsq_buf=None, clip_norm=0.0, ls_dyn=None, ls_good=None, grow_after=200):
    """Advance the device step counter -- unless (sumsq_buf, clip_norm) say that clip_adam_step skipped this step (non-finite norm).
    ls_dyn f32 [m, 1/m] + ls_good int32[1]: the dynamic part of the f16 loss scale follows the outcome (halved by a skipped step, doubled
    back up to 1 after `grow_after` applied steps in a row)."""
    _step_ok(step_dev)
    _req((ls_dyn is None) == (ls_good is None), "step_increment: ls_dyn and ls_good come together")
    if ls_dyn is not None:
        _req(ls_dyn.dtype == torch.float32 and ls_dyn.numel() == 2 and ls_dyn.is_contiguous() and ls_good.dtype == torch.int32 and ls_good.numel() == 1,
             "step_increment: ls_dyn f32[2], ls_good int32[1]")
    call("mnn_step_increment", _stream(), _ptr(step_dev), _ptr(sumsq_buf), float(clip_norm), _ptr(ls_dyn), _ptr(ls_good), int(grow_after))


def bias_grad(dY, db, accumulate=False):
    _rowmajor(dY, "bias_grad dY")
    _req(dY.dtype == torch.float32 and db.dtype == torch.float32 and db.numel() == dY.shape[1] and db.is_contiguous(), "bias_grad: shapes")
    call("mnn_bias_grad", _stream(), _ptr(dY), dY.shape[0], dY.shape[1], dY.stride(0), _ptr(db), int(accumulate))


def grad_rows_fanout(dY, cols_t, out_c, out_t, db):
    """One pass over dY f32 [rows, cols_c]: out_c = bf16 copy, out_t[:cols_t, :rows] = bf16 transpose, db[:cols_t] += column sums."""
    _rowmajor(dY, "fanout dY"); _rowmajor(out_c, "fanout out_c"); _rowmajor(out_t, "fanout out_t")
    rows, cols_c = dY.shape
    _req(dY.dtype == torch.float32 and out_c.dtype in H16 and out_t.dtype == out_c.dtype and db.dtype == torch.float32,
         "fanout: dY/db f32, outputs bf16 / f16")
    _req(out_c.shape == dY.shape and out_t.shape[0] == cols_t and out_t.shape[1] >= rows and 0 < cols_t <= cols_c and db.numel() == cols_t
         and db.is_contiguous(), "fanout: shapes")
    call("mnn_grad_rows_fanout", _stream(), _ptr(dY), rows, cols_c, cols_t, dY.stride(0), _ptr(out_c), out_c.stride(0), _ptr(out_t),
         out_t.stride(0), _ptr(db), dtype_code(out_c))


def fill(x, value):
    _req(x.dtype == torch.float32 and x.is_contiguous(), "fill: f32 contiguous")
    call("mnn_fill_f32", _stream(), _ptr(x), x.numel(), float(value))


# ------------------------------------------------------------------------------------------------
def musical_bar_stats(x, poly_threshold, pattern_class, notes, used_pitches, used_classes, poly_steps, pat_on, pat_tol, beat_chroma):
    """x u8 [B,bars,steps,P,M]; int32 outputs [B*bars, M], beat_chroma int32 [B*bars,4,12,M] (metrics/musical.py)."""
    _req(x.dtype == torch.uint8 and x.dim() == 5 and x.is_contiguous(), "musical_bar_stats: x u8 [B,bars,steps,P,M]")
    B, bars, steps, P, M = x.shape
    for t in (notes, used_pitches, used_classes, poly_steps, pat_on, pat_tol):
        _req(t.dtype == torch.int32 and t.numel() == B * bars * M and t.is_contiguous(), "musical_bar_stats: int32 [B*bars, M] outputs")
    _req(beat_chroma.dtype == torch.int32 and beat_chroma.numel() == B * bars * 48 * M and beat_chroma.is_contiguous(), "musical_bar_stats: beat_chroma")
    _req(pattern_class is None or (pattern_class.dtype == torch.uint8 and pattern_class.numel() == steps), "musical_bar_stats: pattern_class u8 [steps]")
    call("mnn_musical_bar_stats", _stream(), _ptr(x), B * bars, steps, P, M, int(poly_threshold), _ptr(pattern_class), _ptr(notes), _ptr(used_pitches),
         _ptr(used_classes), _ptr(poly_steps), _ptr(pat_on), _ptr(pat_tol), _ptr(beat_chroma))


def musical_note_stats(x, threshold, onsets, qualified):
    """x u8 [B,T,P,M]; onsets / qualified int32 [M], accumulated (zero them first)."""
    _req(x.dtype == torch.uint8 and x.dim() == 4 and x.is_contiguous(), "musical_note_stats: x u8 [B,T,P,M]")
    B, T, P, M = x.shape
    for t in (onsets, qualified):
        _req(t.dtype == torch.int32 and t.numel() == M, "musical_note_stats: int32 [M] outputs")
    call("mnn_musical_note_stats", _stream(), _ptr(x), B, T, P, M, int(threshold), _ptr(onsets), _ptr(qualified))


def eval_counts(targets, predictions, counts):
    """counts int64 [4] += (tp, fp, fn, equal) over all cells of targets / predictions (u8, same shape)."""
    _req(targets.dtype == torch.uint8 and predictions.dtype == torch.uint8 and targets.is_contiguous() and predictions.is_contiguous()
         and targets.numel() == predictions.numel(), "eval_counts: targets / predictions u8, same size")
    _req(counts.dtype == torch.int64 and counts.numel() == 4, "eval_counts: counts int64 [4]")
    call("mnn_eval_counts", _stream(), _ptr(targets), _ptr(predictions), targets.numel(), _ptr(counts))


def log_loss_rows(targets, probs, out):
    """out[row] = sum_d tf.losses.log_loss(targets, probs) (epsilon 1e-7); targets u8 [N,D], probs f32 [N,D] (row stride >= D)."""
    N, D = targets.shape
    _req(targets.dtype == torch.uint8 and targets.is_contiguous(), "log_loss_rows: targets u8 [N,D]")
    _req(probs.dtype == torch.float32 and probs.shape == (N, D) and probs.stride(1) == 1, "log_loss_rows: probs f32 [N,D]")
    _req(out.dtype == torch.float32 and out.numel() == N and out.is_contiguous(), "log_loss_rows: out f32 [N]")
    call("mnn_log_loss_rows", _stream(), _ptr(targets), _ptr(probs), N, D, probs.stride(0), _ptr(out))


# ------------------------------------------------------------------------------------------------
# deterministic f32 single steps of the sampling scan (csrc/det_step.hip)
def lstm_step_det(jobs):
    """One LSTMBlockCell step per job, ALL jobs in one launch.  job = dict(x=[B, >= n_x] u8 | f32 (or None), n_x, x2=f32 [B, n_x2] or None,
    h_prev / c_prev f32 [B, u] or None (zero state), W f32 [(n_x + n_x2 + u), 4u] (TF layout), bias f32 [4u], c_out / h_out f32 [B, u])."""
    _req(1 <= len(jobs) <= 8, "lstm_step_det: 1..8 jobs")
    arr = (_lib.DetLstmJob * len(jobs))()
    B = jobs[0]["c_out"].shape[0]
    for a, j in zip(arr, jobs):
        u = j["c_out"].shape[1]
        x, x2 = j.get("x"), j.get("x2")
        n_x = int(j.get("n_x", x.shape[1] if x is not None else 0))
        n_x2 = x2.shape[1] if x2 is not None else 0
        W, b = j["W"], j["bias"]
        _req(W.dtype == torch.float32 and W.is_contiguous() and tuple(W.shape) == (n_x + n_x2 + u, 4 * u), f"lstm_step_det: W {tuple(W.shape)} != ({n_x + n_x2 + u}, {4 * u})")
        _req(b.dtype == torch.float32 and b.is_contiguous() and b.numel() == 4 * u, "lstm_step_det: bias f32 [4u]")
        for k in ("c_out", "h_out", "h_prev", "c_prev"):
            t = j.get(k)
            _req(t is None or (t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape) == (B, u)), f"lstm_step_det: {k} must be contiguous f32 [B, u]")
        if x is not None:         # any 2-D view: element (row, k) at row * stride(0) + k * stride(1) (one track of a [B, P, M] step has stride(1) = M)
            _req(x.dim() == 2 and x.stride(1) >= 1 and x.shape[0] == B and x.shape[1] >= n_x and x.dtype in (torch.uint8, torch.float32),
                 "lstm_step_det: x is a u8 / f32 [B, >= n_x] view")
        if x2 is not None:
            _rowmajor(x2, "lstm_step_det x2")
            _req(x2.shape[0] == B and x2.dtype == torch.float32, "lstm_step_det: x2 is f32 [B, n_x2]")
        a.x, a.x_dtype, a.n_x, a.ld_x = (_ptr(x) if x is not None else None), (dtype_code(x) if x is not None else F32), n_x, (x.stride(0) if x is not None else 0)
        a.es_x = x.stride(1) if x is not None else 1
        a.x2, a.n_x2, a.ld_x2 = (_ptr(x2) if x2 is not None else None), n_x2, (x2.stride(0) if x2 is not None else 0)
        a.h_prev, a.c_prev = _ptr(j.get("h_prev")), _ptr(j.get("c_prev"))
        a.W, a.bias, a.c_out, a.h_out, a.units = _ptr(W), _ptr(b), _ptr(j["c_out"]), _ptr(j["h_out"]), u
        Wp = j.get("Wp")                                     # det_lstm_pack(W, u): the same numbers in the kernel's load order (optional)
        _req(Wp is None or (Wp.dtype == torch.float32 and Wp.is_contiguous() and Wp.numel() * 4 == det_lstm_pack_bytes(n_x + n_x2 + u, u) and Wp.data_ptr() % 16 == 0),
             "lstm_step_det: Wp is det_lstm_pack(W, units)")
        a.Wp = _ptr(Wp)
    call("mnn_lstm_step_det", _stream(), B, len(jobs), arr)


def det_lstm_pack_bytes(K, units):
    return int(_lib.load().mnn_det_lstm_pack_bytes(int(K), int(units)))


def det_lstm_pack(W, units, out=None):
    """W f32 [K, 4 units] (TF layout) repacked for lstm_step_det's job["Wp"]; to be redone whenever W changes."""
    _req(W.dtype == torch.float32 and W.is_contiguous() and W.dim() == 2 and W.shape[1] == 4 * units and W.shape[0] > units, "det_lstm_pack: W f32 [K, 4 units]")
    n = det_lstm_pack_bytes(W.shape[0], units) // 4
    if out is None:
        out = torch.empty(n, device=W.device, dtype=torch.float32)
    _req(out.dtype == torch.float32 and out.numel() == n and out.is_contiguous() and out.data_ptr() % 16 == 0, "det_lstm_pack: out")
    call("mnn_det_lstm_pack", _stream(), _ptr(W), W.shape[0], int(units), _ptr(out))
    return out


def dense_det(jobs):
    """out = x . W + bias per job (ascending-k fmaf chain), all jobs in one launch.  job = dict(x f32 [B, K], W f32 [K, N] (row pitch >= N),
    bias f32 [N] or None, out f32 [B, >= N] view)."""
    _req(1 <= len(jobs) <= 8, "dense_det: 1..8 jobs")
    arr = (_lib.DetDenseJob * len(jobs))()
    B = jobs[0]["x"].shape[0]
    for a, j in zip(arr, jobs):
        x, W, b, out = j["x"], j["W"], j.get("bias"), j["out"]
        _rowmajor(x, "dense_det x"); _rowmajor(W, "dense_det W"); _rowmajor(out, "dense_det out")
        K, N = W.shape
        _req(x.dtype == W.dtype == out.dtype == torch.float32 and x.shape == (B, K) and out.shape[0] == B and out.shape[1] == N, "dense_det: f32 x [B,K], W [K,N], out [B,N]")
        _req(b is None or (b.dtype == torch.float32 and b.is_contiguous() and b.numel() == N), "dense_det: bias f32 [N]")
        a.x, a.ld_x, a.K, a.W, a.ld_w, a.N, a.bias, a.out, a.ld_out = _ptr(x), x.stride(0), K, _ptr(W), W.stride(0), N, _ptr(b), _ptr(out), out.stride(0)
        Wp = j.get("Wp")                                     # det_dense_pack(W): the same numbers in the kernel's load order (optional)
        _req(Wp is None or (Wp.dtype == torch.float32 and Wp.is_contiguous() and Wp.numel() * 4 == int(_lib.load().mnn_det_dense_pack_bytes(K, N)) and Wp.data_ptr() % 16 == 0),
             "dense_det: Wp is det_dense_pack(W)")
        a.Wp = _ptr(Wp)
    call("mnn_dense_det", _stream(), B, len(jobs), arr)


def det_dense_pack(W, out=None):
    """W f32 [K, N] (row pitch >= N) repacked for dense_det's job["Wp"]; to be redone whenever W changes."""
    _rowmajor(W, "det_dense_pack W")
    _req(W.dtype == torch.float32 and W.dim() == 2, "det_dense_pack: W f32 [K, N]")
    K, N = W.shape
    n = int(_lib.load().mnn_det_dense_pack_bytes(K, N)) // 4
    if out is None:
        out = torch.empty(n, device=W.device, dtype=torch.float32)
    _req(out.dtype == torch.float32 and out.numel() == n and out.is_contiguous() and out.data_ptr() % 16 == 0, "det_dense_pack: out")
    call("mnn_det_dense_pack", _stream(), _ptr(W), K, N, W.stride(0), _ptr(out))
    return out


def generate_scan(intro, num_steps, layers, dense_W, dense_bias, tracks, D, Hn, w_enc, w_dec, temperature, seed, row0):
    """mnn_generate_scan: the whole sampling scan of an LSTM-(Multi)NADE generator in one call.  intro u8 [B, Ti, tracks * D]; layers = [(W, bias)]
    f32 master weights; returns samples u8 [B, num_steps, tracks * D]."""
    _req(intro.dtype == torch.uint8 and intro.dim() == 3 and intro.is_contiguous() and intro.shape[2] == tracks * D, "generate_scan: intro u8 [B, Ti, tracks * D]")
    B, Ti, n_in = intro.shape
    n_out = tracks * (Hn + D)
    arr = (_lib.ScanLstmLayer * len(layers))()
    k_in = n_in
    for a, (W, b) in zip(arr, layers):
        u = b.numel() // 4
        _req(W.dtype == torch.float32 and W.is_contiguous() and tuple(W.shape) == (k_in + u, 4 * u) and b.dtype == torch.float32 and b.is_contiguous(),
             "generate_scan: layer weights f32 [(n_in + u), 4u] / [4u]")
        a.W, a.bias, a.units = _ptr(W), _ptr(b), u
        k_in = u
    _req(dense_W.dtype == torch.float32 and dense_W.is_contiguous() and tuple(dense_W.shape) == (k_in, n_out), "generate_scan: Dense kernel f32 [units_last, n_out]")
    _req(dense_bias is None or (dense_bias.dtype == torch.float32 and dense_bias.is_contiguous() and dense_bias.numel() == n_out), "generate_scan: Dense bias")
    for w in (w_enc, w_dec):
        _req(w.dtype == torch.float32 and w.is_contiguous() and w.numel() == tracks * D * Hn, "generate_scan: NADE weights f32 [tracks, D, Hn]")
    need = int(_lib.load().mnn_generate_scan_workspace_bytes(B, n_in, len(layers), arr, n_out))
    ws = torch.empty(need + 256, dtype=torch.uint8, device=intro.device)
    off = (-ws.data_ptr()) % 256
    samples = torch.empty((B, int(num_steps), n_in), dtype=torch.uint8, device=intro.device)
    call("mnn_generate_scan", _stream(), B, Ti, int(num_steps), _ptr(intro), n_in, len(layers), arr, _ptr(dense_W), _ptr(dense_bias), n_out, tracks, D, Hn,
         _ptr(w_enc), _ptr(w_dec), float(-1.0 if temperature is None else temperature), int(seed), int(row0), _ptr(samples),
         C.c_void_p(ws.data_ptr() + off), need)
    return samples                          # (the workspace returns to the allocator in stream order: later users are behind the scan)
