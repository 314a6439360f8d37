"""Shared by the attention parity tests: operands for the kernels' pre-scaled-q contract (tad_attn_fwd / tad_attn_bwd with
q_prescaled != 0: the q third of qkv carries scale * log2(e), as tad_linear_fwd_qkv's q_prescale writes it)."""
LOG2E = 1.4426950408889634


def prescaled_pair(qkv, B, N, H, scale, rnd, d=64):
    """qkv: [B*N, 3*H*d] float32 whose values are exact in the operand format; rnd: round-trip through that format.
    Returns (kernel operand: q third replaced by rnd(q * scale * log2e); oracle input in float64: q third = that operand's q divided
    by scale * log2e) -- the kernels and the oracle then see the SAME q, k, v, and the kernel's dq is the gradient of that plain q."""
    c = scale * LOG2E
    q5 = qkv.double().reshape(B, N, 3, H, d).clone()
    qp = rnd((q5[:, :, 0] * c).float())
    opnd = qkv.reshape(B, N, 3, H, d).clone()
    opnd[:, :, 0] = qp
    q5[:, :, 0] = qp.double() / c
    return opnd.reshape(B * N, -1), q5.reshape(B * N, -1)
