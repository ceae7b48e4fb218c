"""Host-side (PyTorch) quantisation arithmetic shared by the pack step and the nemo façade.

These are the one-time, per-model computations the reference redoes on every forward
(SURVEY §2.3 K12): BN folding, per-channel weight / 32-bit bias integers and the
fixed-point requantisation multipliers.  Float32 operation order follows the reference
exactly so the integers are bit-identical to what its modules compute:
  scale      symmetric_linear_quantization_params  nemo/quantization/utils/quant_utils.py:28-54
  quantise   linear_quantize + SymmetricQuantFunction            quant_utils.py:12-26,57-79
  fold       QuantConv1d.forward (BN folded, fixed)              quant_modules.py:352-364
  (m, e)     batch_frexp / fixedpoint_mul                        quant_utils.py:121-147,187-196
"""
import torch


def qrange(bits: int):
    """Clamp range of fixedpoint_mul in symmetric mode: [-n-1, n], n = 2^(bits-1)-1."""
    n = 2 ** (bits - 1) - 1
    return -n - 1, n


# Scale arithmetic (divisions, frexp/ldexp) always runs on the host CPU on the tiny per-tensor / per-channel
# vectors and only the element-wise multiply / round / clamp runs on the tensor's device: IEEE multiply is
# identical everywhere, while GPU float division, pow and ldexp in PyTorch-ROCm are not guaranteed correctly
# rounded - a 1-ulp scale difference would make host-calibrated integers differ from the packed engine's.
def sym_scale(bits: int, lo: torch.Tensor, hi: torch.Tensor) -> torch.Tensor:
    n = 2 ** (bits - 1) - 1
    dev = lo.device
    m = torch.maximum(lo.detach().float().cpu().abs(), hi.detach().float().cpu().abs())
    return (torch.clamp(m, min=1e-8) / n).to(dev)


def quantize(x: torch.Tensor, bits: int, scale: torch.Tensor) -> torch.Tensor:
    """Integers (as float32, like the reference) in [-n, n-1]."""
    n = 2 ** (bits - 1) - 1
    inv = (1.0 / scale.detach().float().cpu()).to(x.device)
    return torch.clamp(torch.round(inv * x), -n, n - 1)


def ieee_sqrt(x: torch.Tensor) -> torch.Tensor:
    """Correctly rounded float32 sqrt.  torch's CPU sqrt (MKL VML) is 1 ulp off for ~0.7 % of
    inputs and differs between host CPUs, which would make packed weights host-dependent;
    float32(sqrt(float64(x))) is the IEEE result (and what a GPU run of the reference computes)."""
    import numpy as np
    return torch.from_numpy(np.sqrt(x.detach().cpu().numpy().astype(np.float64)).astype(np.float32)).to(x.device)


def fold_bn(weight, bias, gamma, beta, mean, var, eps=1e-3):
    dev = weight.device
    std = ieee_sqrt(var.detach().float().cpu() + eps)
    g = (gamma.detach().float().cpu() / std).to(dev)
    w = weight * g.reshape(-1, 1, 1)
    b = torch.zeros_like(mean) if bias is None else bias
    return w, (b - mean) * g + beta


def weight_integers(weight: torch.Tensor, wbit: int):
    """-> (W_int float32 tensor, s_w [Cout])."""
    flat = weight.reshape(weight.shape[0], -1)
    s_w = sym_scale(wbit, flat.min(dim=1).values, flat.max(dim=1).values)
    return quantize(weight, wbit, s_w.view(-1, 1, 1)), s_w


def bias_integers(bias, s_w: torch.Tensor, s_x: torch.Tensor, bits: int = 32):
    """-> (B_int float32 tensor or None, s_b [Cout]); `bits`-wide range, float32 evaluation."""
    s_b = s_w * s_x.reshape(-1)[0]
    if bias is None:
        return None, s_b
    return quantize(bias, bits, s_b), s_b


def requant_multiplier(pre_sf: torch.Tensor, out_sf: torch.Tensor) -> torch.Tensor:
    """M = m * 2^-e (float64) with (m, e) = batch_frexp(f64(pre_sf) / f64(f32(out_sf))).

    rint(f64(z) * M) equals fixedpoint_mul's rint(f64(z) * f64(m) / 2^e) bit for bit:
    scaling by a power of two commutes with the fp64 rounding of the product."""
    import numpy as np
    r = pre_sf.detach().float().cpu().double().numpy() / out_sf.detach().float().cpu().double().numpy()
    mant, ex = np.frexp(r)
    m = np.floor(mant * float(2 ** 31) + 0.5)               # Decimal ROUND_HALF_UP on a positive value
    return torch.from_numpy(np.ldexp(m, ex - 31)).to(pre_sf.device)


def requant(z: torch.Tensor, M: torch.Tensor, lo: int, hi: int) -> torch.Tensor:
    """Integer requantisation of int-valued tensor z [B,C,T] with per-channel or scalar M (float64)."""
    return torch.clamp(torch.round(z.double() * M.reshape(1, -1, 1)), lo, hi)
