"""Zero-shot calibration data (behaviour of nemo/quantization/utils/distill_data.py:11-162).

What it computes: mel-domain batches x [batch, dim, seqlen], started from uniform noise in [-0.3, 0.3], moved by Adam so
that, in the FLOAT model (quant mode 'none', BatchNorm layers not folded), the per-channel mean / standard deviation of
every conv output that feeds a BatchNorm matches that BatchNorm's running statistics:

    loss(x) = sum over (conv, bn) of  mean_c KL( N(bn.mean_c, bn.std_c) || N(mean_c(conv(x)), std_c(conv(x))) )

with std = sqrt(var + 1e-6) on both sides, statistics over the batch and time axes, one Adam(lr) + ReduceLROnPlateau
(min_lr 1e-4, patience 25) per batch and `train_iter` steps.  The result is what `synthesize.py` dumps and
`inference.py --load` calibrates on.

Adjacent to the hot path, not part of it: fp32 forward + backward in plain PyTorch-ROCm on whatever device the model
lives on.  Pinned by tests/golden/distill.npz, which tests/golden/gen_golden.py produced by running the reference's own
module for 3 iterations from a fixed start (tests/test_facade_cpu.py::test_distill_matches_reference_fixture).
Beyond the reference's signature: `seed` (reproducible noise), `init` (caller-supplied start batches), `history`
(receives every iteration's loss), `verbose`."""
import contextlib

import torch

EPS = 1e-6
NOISE_AMPLITUDE = 0.3                                         # the start is U(-0.3, 0.3) (distill_data.py:11-25)


def uniform_batches(num_batch, batch_size, dim, seqlen, seed=None):
    """`num_batch` tensors [batch_size, dim, seqlen] of uniform noise in [-0.3, 0.3] (the reference draws them through a
    10000-sample Dataset and a 32-worker DataLoader; one generator call per batch gives the same distribution)."""
    gen = None if seed is None else torch.Generator().manual_seed(int(seed))
    for _ in range(num_batch):
        yield (2.0 * torch.rand(batch_size, dim, seqlen, generator=gen) - 1.0) * NOISE_AMPLITUDE


def gaussian_kl(mean_p, std_p, mean_q, std_q):
    """Channel-mean of KL(N(mean_p, std_p) || N(mean_q, std_q)) = log(std_q / std_p) + (std_p^2 + (mean_p - mean_q)^2) / (2 std_q^2) - 1/2
    (distill_data.py:59-68 writes the same quantity as a - b with b = (1 - c) / 2)."""
    ratio = (std_p.square() + (mean_p - mean_q).square()) / std_q.square()
    return (torch.log(std_q / std_p) + 0.5 * ratio - 0.5).mean()


_kl_loss = gaussian_kl                                       # the reference's name for it


class _BatchNormTarget:
    """One (conv, BatchNorm) pair: keeps the conv's latest output (forward hook) and scores its statistics against the
    BatchNorm's running mean / variance."""

    def __init__(self, conv, bn, device):
        if not isinstance(bn, torch.nn.BatchNorm1d):
            raise TypeError(f'convs_before_bn pairs a conv with {type(bn).__name__}, expected BatchNorm1d')
        self.mean = bn.running_mean.detach().flatten().to(device).clone()
        self.std = (bn.running_var.detach().flatten().to(device) + EPS).sqrt()
        self.output = None
        self._handle = conv.register_forward_hook(self._capture)

    def _capture(self, module, inputs, output):
        self.output = output[0] if isinstance(output, (tuple, list)) else output    # MaskedConv1d returns (y, lens, scale)

    def release(self):
        self._handle.remove()
        self.output = None

    def score(self):
        y = self.output
        if y is None:
            raise RuntimeError('a conv listed in convs_before_bn did not run in the forward pass')
        mean = y.mean(dim=(0, 2))
        std = (y.var(dim=(0, 2)) + EPS).sqrt()
        if mean.shape != self.mean.shape:
            raise ValueError(f'conv output has {tuple(mean.shape)} channels, its BatchNorm {tuple(self.mean.shape)}')
        return gaussian_kl(self.mean, self.std, mean, std)


@contextlib.contextmanager
def _batchnorm_targets(encoder, device):
    targets = [_BatchNormTarget(conv, bn, device) for conv, bn in encoder.convs_before_bn]
    try:
        yield targets
    finally:
        for t in targets:
            t.release()


@contextlib.contextmanager
def _frozen(*modules):
    """Gradients flow to the INPUT only: the weights' requires_grad is switched off for the duration (the reference
    zeroes their accumulated gradients every step instead; the input's gradient is the same)."""
    params = [p for m in modules for p in m.parameters() if p.requires_grad]
    for p in params:
        p.requires_grad_(False)
    try:
        yield
    finally:
        for p in params:
            p.requires_grad_(True)


def get_synthetic_data(teacher_model, teacher_model_decoder, batch_size, dim, seqlen, train_iter=500, num_batch=1, lr=0.01,
                       seed=None, verbose=True, history=None, init=None):
    """-> list of `num_batch` tensors [batch_size, dim, seqlen] on the model's device (detached).
    teacher_model: the float encoder (`convs_before_bn` = its (conv, BatchNorm1d) pairs), teacher_model_decoder: its
    decoder (run for parity with the reference's forward; it does not enter the loss)."""
    device = next(teacher_model.parameters()).device
    teacher_model.eval()
    teacher_model_decoder.eval()
    starts = list(init) if init is not None else uniform_batches(num_batch, batch_size, dim, seqlen, seed)
    lengths = torch.full((batch_size,), seqlen, dtype=torch.long, device=device)
    refined = []
    with torch.enable_grad(), _frozen(teacher_model, teacher_model_decoder), _batchnorm_targets(teacher_model, device) as targets:
        for n, start in enumerate(starts):
            if n == num_batch:
                break
            if verbose:
                print(f'Distillation: {n + 1} / {num_batch}')
            x = start.detach().to(device=device, dtype=torch.float32).clone().requires_grad_(True)
            optimizer = torch.optim.Adam([x], lr=lr)
            plateau = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, min_lr=1e-4, patience=25)
            for _ in range(train_iter):
                optimizer.zero_grad(set_to_none=True)
                encoded, _, encoded_scale = teacher_model(x, lengths)
                teacher_model_decoder(encoder_output=encoded, encoder_output_scaling_factor=encoded_scale)
                loss = torch.stack([t.score() for t in targets]).sum()
                loss.backward()
                optimizer.step()
                value = float(loss.detach())
                plateau.step(value)
                if history is not None:
                    history.append(value)
            refined.append(x.detach().clone())
    return refined
