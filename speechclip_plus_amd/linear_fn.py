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
        wb = weight.detach().to(torch.bfloat16).contiguous()
        y = ops.linear_bf16(xb, wb, None if bias is None else bias.detach().float().contiguous())
        ctx.save_for_backward(xb, wb)
        ctx.meta = (shape, rows, rp, K, N, x.dtype, bias is not None)
        return y[:rows].to(x.dtype).reshape(*shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        xb, wb = ctx.saved_tensors
        shape, rows, rp, K, N, dtype, has_bias = ctx.meta
        dyb = torch.zeros(rp, N, device=dy.device, dtype=torch.bfloat16)
        dyb[:rows] = dy.reshape(rows, N)
        dx = gW = gb = None
        if ctx.needs_input_grad[0]:
            dx = ops.linear_bf16(dyb, wb.t().contiguous())[:rows].to(dtype).reshape(shape)
        if ctx.needs_input_grad[1]:
            gW = torch.empty(N, K, device=dy.device, dtype=torch.float32)
            if has_bias and ctx.needs_input_grad[2]:
                gb = torch.empty(N, device=dy.device, dtype=torch.float32)
            ops.wgrad_bf16(dyb, xb, gW, gb, beta=0.0)
        elif has_bias and ctx.needs_input_grad[2]:
            gb = dy.reshape(rows, N).float().sum(0)
        return dx, gW, gb


def linear_bf16_autograd(x: torch.Tensor, weight: torch.Tensor, bias=None) -> torch.Tensor:
    """Drop-in for ``F.linear`` on a GPU when K and N are multiples of 64 / 8; plain ``F.linear`` otherwise (CPU, odd shapes)."""
    if x.is_cuda and x.shape[-1] % 64 == 0 and weight.shape[0] % 8 == 0:
        return LinearBf16Fn.apply(x, weight, bias)
    return torch.nn.functional.linear(x, weight, bias)
