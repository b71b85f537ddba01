"""fp8 (OCP e4m3fn) operands with one power-of-two scale per row -- the operand format of s6d_gemm_fp8 (BASELINE configs[4]).

A row r of a matrix is stored as bytes q[r, :] (e4m3fn) and one E8M0 byte s[r]: value = q * 2^(s - 127).  The exponent is the
smallest e for which amax(row) / 2^e <= 448 (e4m3's largest finite value), so the row uses the top binade of the format; the
scales are powers of two because they ride in the hardware block-scale operands of the gfx950 matrix instruction
(v_mfma_scale_f32_32x32x64_f8f6f4) instead of costing an epilogue multiply.  Weights are quantised once per weight version
(per output channel = per row of nn.Linear.weight) with the library statements below; activations by the LayerNorm kernel
(s6d_layernorm_fp8), which follows the same rule."""
import torch

E4M3_MAX = 448.0


def row_exponents(x):
    """(rows, K) float -> (rows,) int32 exponent e: the smallest with amax / 2^e <= 448 (0 for an all-zero row)."""
    amax = x.abs().amax(dim=1).float()
    f, ex = torch.frexp(amax)                                  # amax = f 2^ex, f in [0.5, 1)
    e = ex - torch.where(f <= 0.875, 9, 8)                      # 512 f <= 448 iff f <= 0.875
    return torch.where(amax > 0, e, torch.zeros_like(e)).clamp(-127, 127).to(torch.int32)


def quantize_rows(x):
    """(rows, K) float -> (q (rows, K) uint8 = e4m3fn bytes, s (rows,) uint8 = E8M0 scale bytes)."""
    e = row_exponents(x)
    y = torch.ldexp(x.float(), -e[:, None])
    q = y.clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn)
    return q.view(torch.uint8).contiguous(), (e + 127).to(torch.uint8).contiguous()


def dequantize_rows(q, s):
    """Inverse of quantize_rows, in float32."""
    return torch.ldexp(q.view(torch.float8_e4m3fn).float(), (s.to(torch.int32) - 127)[:, None])


def quantize_blocks(x, block=32):
    """MX form: (rows, K) float -> (q (rows, K) uint8 e4m3fn bytes, s (rows, K / block) uint8 E8M0 bytes), one power-of-two scale
    per row and ``block`` consecutive elements, by the rule of quantize_rows applied per block (what the GELU epilogue of
    s6d_gemm_fp8_gelu_mx does for every 32 columns of a row)."""
    rows, K = x.shape
    q, s = quantize_rows(x.reshape(rows * (K // block), block))
    return q.view(rows, K), s.view(rows, K // block)


def dequantize_blocks(q, s, block=32):
    """Inverse of quantize_blocks, in float32."""
    rows, K = q.shape
    return dequantize_rows(q.reshape(rows * (K // block), block), s.reshape(-1)).view(rows, K)


def cached_weight(lin, w2d=None):
    """(q, s, bias_f32) of an nn.Linear, cached until the parameters change."""
    w = lin.weight if w2d is None else w2d
    b = lin.bias
    key = (w._version, w.data_ptr(), None if b is None else (b._version, b.data_ptr()))
    c = getattr(lin, "_s6d_fp8", None)
    if c is None or c[0] != key:
        q, s = quantize_rows(w.detach().float())
        c = (key, q, s, None if b is None else b.detach().float().contiguous())
        lin._s6d_fp8 = c
    return c[1], c[2], c[3]
