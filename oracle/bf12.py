"""CPU restatement (numpy) of the lossless 12-bit weight format of the decode linears (include/ivlm_hip.h: ivlm_pack_bf12m_count /
_fill, ivlm_gemv1_bf12m).  TEST INFRASTRUCTURE ONLY: imported by tests/, never by the product path.

The format has no counterpart in the reference (it is a storage form of the checkpoint's own bf16 weights, model/InteractVLM.py:524-531
being the decode loop that streams them); what this file pins is that the packing is LOSSLESS and what the plane layout is.

    row r:  ebase[r] = max(0, max exponent field of the row - 15)
    weight: code = exponent field - ebase[r]; in the window (code 1 .. 15): P byte = sign << 7 | mantissa (7 bits), E nibble = code;
            otherwise P = 0, code = 0 and, if the weight is nonzero, a patch (column, bf16 bits) of row r, in column order
    fragment layout (n_rows % 16 == 0, K % 64 == 0), byte offsets:
            P[((((rb * K/64 + sp) * 4 + q) * 16 + r) * 2 + h) * 8 + i]     = P byte of weight (rb * 16 + r, sp * 64 + h * 32 + q * 8 + i)
            E[((((rb * K/64 + sp) * 4 + q) * 16 + r) * 2 + h) * 4 + i // 2]: low nibble = even i, high nibble = odd i
"""
import numpy as np


def pack(w_bits: np.ndarray, n_rows: int = None):
    """w_bits uint16 [N, K] (bf16 bit patterns) -> dict(P, E (fragment layout, flat uint8), ebase, patch_ptr, patch_col, patch_val)."""
    w_bits = np.asarray(w_bits, dtype=np.uint16)
    N, K = w_bits.shape
    n_rows = N if n_rows is None else n_rows
    assert n_rows % 16 == 0 and n_rows >= N and K % 64 == 0
    b = np.zeros((n_rows, K), dtype=np.int64)
    b[:N] = w_bits
    ef = (b >> 7) & 0xFF
    ebase = np.maximum(ef.max(axis=1) - 15, 0)
    code = ef - ebase[:, None]
    inwin = code >= 1
    esc = (~inwin) & ((b & 0x7FFF) != 0)
    P = np.where(inwin, ((b >> 8) & 0x80) | (b & 0x7F), 0).astype(np.uint8)
    code = np.where(inwin, code, 0)
    E = (code[:, 0::2] | (code[:, 1::2] << 4)).astype(np.uint8)
    nsp = K // 64
    Pf = P.reshape(n_rows // 16, 16, nsp, 2, 4, 8).transpose(0, 2, 4, 1, 3, 5).reshape(-1)
    Ef = E.reshape(n_rows // 16, 16, nsp, 2, 4, 4).transpose(0, 2, 4, 1, 3, 5).reshape(-1)
    rows, cols = np.nonzero(esc)  # row-major: sorted by row, then column
    ptr = np.zeros(n_rows + 1, dtype=np.int32)
    ptr[1:] = np.cumsum(np.bincount(rows, minlength=n_rows))
    return dict(P=Pf.copy(), E=Ef.copy(), ebase=ebase.astype(np.int32), patch_ptr=ptr, patch_col=cols.astype(np.int32),
                patch_val=b[rows, cols].astype(np.uint16), shape=(n_rows, K))


def unpack(p):
    """-> uint16 [n_rows, K]: the bf16 bit patterns that were packed (-0.0 comes back as +0.0)."""
    n_rows, K = p["shape"]
    nsp = K // 64
    P = p["P"].reshape(n_rows // 16, nsp, 4, 16, 2, 8).transpose(0, 3, 1, 4, 2, 5).reshape(n_rows, K).astype(np.int64)
    E = p["E"].reshape(n_rows // 16, nsp, 4, 16, 2, 4).transpose(0, 3, 1, 4, 2, 5).reshape(n_rows, K // 2).astype(np.int64)
    code = np.empty((n_rows, K), dtype=np.int64)
    code[:, 0::2] = E & 0xF
    code[:, 1::2] = E >> 4
    out = np.where(code > 0, ((P & 0x80) << 8) | ((p["ebase"][:, None].astype(np.int64) + code) << 7) | (P & 0x7F), 0)
    for r in range(n_rows):
        s, e = int(p["patch_ptr"][r]), int(p["patch_ptr"][r + 1])
        out[r, p["patch_col"][s:e]] = p["patch_val"][s:e]
    return out.astype(np.uint16)


def gemv(p, x: np.ndarray):
    """y = W x in float64 from the packed planes (the arithmetic the kernels restate: exact bf16 weight values)."""
    w = unpack(p).astype(np.uint32) << 16
    return w.view(np.float32).astype(np.float64) @ np.asarray(x, dtype=np.float64)
