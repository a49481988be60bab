"""``y = x W^T + b`` with bf16 operands and fp32 accumulation on the library's GEMM, as an autograd function for the trainable
projections of the cascaded+/hybrid+ tails (the branch's attention block in_proj / out_proj over B x 499 frames: 113 / 38
GFLOP each way, which stock fp32 GEMMs run at ~100 TFLOP/s).  The reference trains these under ``precision: 16`` autocast
(config/speechCLIP+/*: trainer.precision 16), i.e. in half precision with fp32 accumulation.

forward   y  = sc_gemm_bf16(x, W) + b                (bias in the GEMM epilogue)
backward  dx = sc_gemm_bf16(dy, W^T)                  (transposed bf16 copy of the weight)
          dW = dy^T x  (ops.wgrad_bf16: bf16 transposes + split-K batched GEMM, fp32 partials) ;  db = column sums of dy
Rows are padded to a multiple of 64 (zero rows) for the weight-gradient product.
"""
import torch

from . import ops


def _wgrad_into(params, dyb, xb, N, K, want_bias):
    """dW = dy^T x (+ db = column sums of dy) by ops.wgrad_bf16: added straight into the optimiser's gradient buffers where the
    parameters have them (ops.grad_target; the node then returns None for them: no temporary, no AccumulateGrad add launch), else
    into fresh tensors.  -> (gW, gb) to return from backward."""
    weight, bias = params
    tW = ops.grad_target(weight) if weight.dim() == 2 else None
    tb = ops.grad_target(bias) if (want_bias and bias is not None) else None
    if tW is not None and (tb is not None or not want_bias):
        ops.wgrad_bf16(dyb, xb, tW, tb, beta=1.0)
        return None, None
    gW = torch.empty(N, K, device=dyb.device, dtype=torch.float32)
    gb = torch.empty(N, device=dyb.device, dtype=torch.float32) if want_bias else None
    ops.wgrad_bf16(dyb, xb, gW, gb, beta=0.0)
    return gW, gb


class LinearBf16Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        shape = x.shape
        K = shape[-1]
        N = weight.shape[0]
        rows = x.numel() // K
        rp = (rows + 63) // 64 * 64
        xb = torch.zeros(rp, K, device=x.device, dtype=torch.bfloat16)
        xb[:rows] = x.reshape(rows, K)
        if weight.dim() == 2:
            wb, ctx.wT = ops.derived_pair(weight)                                         # once per parameter version, one launch
        else:
            wb, ctx.wT = ops.derived(weight, "bf16", lambda t: t.to(torch.bfloat16).contiguous()), None
        y = ops.linear_bf16(xb, wb, None if bias is None else bias.detach().float().contiguous())
        ctx.save_for_backward(xb, wb)
        ctx.meta = (shape, rows, rp, K, N, x.dtype, bias is not None)
        ctx.params = (weight, bias)
        return y[:rows].to(x.dtype).reshape(*shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        xb, wb = ctx.saved_tensors
        shape, rows, rp, K, N, dtype, has_bias = ctx.meta
        dyb = torch.zeros(rp, N, device=dy.device, dtype=torch.bfloat16)
        dyb[:rows] = dy.reshape(rows, N)
        dx = gW = gb = None
        if ctx.needs_input_grad[0]:
            dx = ops.linear_bf16(dyb, ctx.wT if ctx.wT is not None else wb.t().contiguous())[:rows].to(dtype).reshape(shape)
        if ctx.needs_input_grad[1]:
            gW, gb = _wgrad_into(ctx.params, dyb, xb, N, K, has_bias and ctx.needs_input_grad[2])
        elif has_bias and ctx.needs_input_grad[2]:
            gb = dy.reshape(rows, N).float().sum(0)
        return dx, gW, gb


class LinearF32Fn(torch.autograd.Function):
    """``y = x W^T + b`` in exact fp32 on the matrix pipe (sc_sgemm_mfma_f32), forward and backward, no transposed copies: the
    three products read x, W and dy in place, each either row-major or K-major.  Used where a rounding of the operands would
    change a DISCRETE result downstream (inference-time CIF: the keyword count is floor(sum alpha); the keyword projection in front of
    the vocabulary argmax).  ``bf16_backward``: only the FORWARD decides something discrete - the two gradient products then run
    on the bf16 GEMMs like every other trainable projection (the reference trains them under precision-16 autocast): 10 + 20 us
    instead of 2 x 70-120 us for the 1600-row products of the keyword projection."""

    @staticmethod
    def forward(ctx, x, weight, bias, bf16_backward=False):
        shape = x.shape
        K, N = shape[-1], weight.shape[0]
        x2 = x.detach().reshape(-1, K).float().contiguous()
        w = weight.detach().float().contiguous()
        y = ops.sgemm_mfma(x2, w, bias=None if bias is None else bias.detach().float().contiguous())
        rows = x2.shape[0]
        ctx.fast = bool(bf16_backward) and K % 256 == 0 and N % 256 == 0 and rows >= 64
        ctx.save_for_backward(x2, w)
        ctx.meta = (shape, x.dtype, bias is not None)
        ctx.params = (weight, bias)
        return y.reshape(*shape[:-1], N).to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        shape, dtype, has_bias = ctx.meta
        N, K = w.shape
        dy2 = dy.reshape(-1, N).float().contiguous()
        rows = dy2.shape[0]
        dx = gW = gb = None
        if ctx.fast:
            rp = (rows + 63) // 64 * 64

            def rows64(t, width):              # bf16 rows, zero-padded to a multiple of 64 (one cast when nothing is to pad)
                if rp == rows:
                    return t.to(torch.bfloat16)
                b = torch.zeros(rp, width, device=dy.device, dtype=torch.bfloat16)
                b[:rows] = t
                return b

            dyb = rows64(dy2, N)
            if ctx.needs_input_grad[0]:
                dx = ops.linear_bf16(dyb, ops.weight_copies(w, "T")[1])[:rows].to(dtype).reshape(shape)
            if ctx.needs_input_grad[1]:
                gW, gb = _wgrad_into(ctx.params, dyb, rows64(x2, K), N, K, has_bias and ctx.needs_input_grad[2])
            elif has_bias and ctx.needs_input_grad[2]:
                gb = dy2.sum(0)
            return dx, gW, gb, None
        if ctx.needs_input_grad[0]:
            dx = ops.sgemm_mfma(dy2, w, b_kmajor=True).to(dtype).reshape(shape)        # dx[m, k] = sum_n dy[m, n] W[n, k]
        if ctx.needs_input_grad[1]:
            gW = ops.sgemm_mfma(dy2, x2, a_kmajor=True, b_kmajor=True)                 # gW[n, k] = sum_m dy[m, n] x[m, k]
        if has_bias and ctx.needs_input_grad[2]:
            gb = torch.empty(N, device=dy.device, dtype=torch.float32)
            ops.colsum(dy2, N, rows, N, gb)
        return dx, gW, gb, None


def linear_f32_autograd(x: torch.Tensor, weight: torch.Tensor, bias=None, bf16_backward: bool = False) -> torch.Tensor:
    if not x.is_cuda:
        raise RuntimeError("speechclip_plus_amd linear layers run on the HIP kernels: device tensors only")
    return LinearF32Fn.apply(x, weight, bias, bf16_backward)


def linear_bf16_autograd(x: torch.Tensor, weight: torch.Tensor, bias=None) -> torch.Tensor:
    """``F.linear`` on the library's bf16 MFMA GEMM (K a multiple of 64, N of 8)."""
    if not x.is_cuda:
        raise RuntimeError("speechclip_plus_amd linear layers run on the HIP kernels: device tensors only")
    if x.shape[-1] % 64 != 0 or weight.shape[0] % 8 != 0:
        raise NotImplementedError(f"linear_bf16_autograd: K = {x.shape[-1]} must be a multiple of 64 and N = {weight.shape[0]} of 8")
    return LinearBf16Fn.apply(x, weight, bias)
