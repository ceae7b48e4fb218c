"""Zero-shot calibration data: synthetic mel-domain batches whose intermediate activations match the BatchNorm
statistics of the pretrained float model (nemo/quantization/utils/distill_data.py:11-162).

Adjacent to the hot path, not part of it: fp32 forward + backward of the float ('none' mode) encoder in plain
PyTorch-ROCm, on whatever device the model lives on (the reference hard-codes .cuda()).  Same function names, argument
meaning, loss and optimiser schedule; the data loader runs in-process (the reference spawns 32 workers to draw uniform
noise) and takes an optional seed so that runs are reproducible."""
import torch
import torch.nn as nn
import torch.optim as optim
from torch.utils.data import DataLoader, Dataset


class UniformDataset(Dataset):
    """Random uniform samples from [-0.3, 0.3] (distill_data.py:11-25)."""

    def __init__(self, length, size, transform=None, seed=None):
        self.length, self.size, self.transform = length, size, transform
        self.gen = None if seed is None else torch.Generator().manual_seed(seed)

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
        return torch.rand(self.size, generator=self.gen) * 0.6 - 0.3


class OutputHook(object):
    """Forward hook keeping the output of an intermediate layer (distill_data.py:27-39)."""

    def __init__(self):
        self.outputs = None

    def hook(self, module, input, output):
        self.outputs = output

    def clear(self):
        self.outputs = None


def _get_random_data(batch_size=32, dim=64, seqlen=500, seed=None):
    """Data loader of uniform samples [batch_size, dim, seqlen] (distill_data.py:41-57)."""
    return DataLoader(UniformDataset(length=10000, size=(dim, seqlen), seed=seed), batch_size=batch_size, shuffle=False,
                      num_workers=0)


def _kl_loss(bn_mean, bn_std, tmp_mean, tmp_std):
    """KL divergence between the Gaussians (bn_mean, bn_std) and (tmp_mean, tmp_std) (distill_data.py:59-68)."""
    a = torch.log(tmp_std / bn_std)
    c = (bn_std ** 2 + (bn_mean - tmp_mean) ** 2) / tmp_std ** 2
    b = 0.5 * (1 - c)
    return (a - b).mean()


def get_synthetic_data(teacher_model, teacher_model_decoder, batch_size, dim, seqlen, train_iter=500, num_batch=1,
                       lr=0.01, seed=None, verbose=True, history=None):
    """distill_data.py:71-162.  `teacher_model`: the float encoder (quant mode 'none', BatchNorm layers NOT folded: the
    hooks sit on every conv that feeds a BatchNorm, `convs_before_bn`), `teacher_model_decoder`: its decoder.
    Returns a list of `num_batch` tensors [batch_size, dim, seqlen] on the model's device.  `history` (optional list)
    receives the loss of every iteration."""
    dataloader = _get_random_data(batch_size, dim, seqlen, seed)
    eps = 1e-6
    device = next(teacher_model.parameters()).device
    teacher_model = teacher_model.eval()
    hooks, hook_handles, bn_stats, refined_gaussian = [], [], [], []
    for conv, bn in teacher_model.convs_before_bn:
        assert isinstance(bn, nn.BatchNorm1d)
        hook = OutputHook()
        hooks.append(hook)
        hook_handles.append(conv.register_forward_hook(hook.hook))
        bn_stats.append((bn.running_mean.detach().clone().flatten().to(device),
                         torch.sqrt(bn.running_var + eps).detach().clone().flatten().to(device)))
    assert len(hooks) == len(bn_stats)
    was_enabled = torch.is_grad_enabled()
    torch.set_grad_enabled(True)
    try:
        for i, gaussian_data in enumerate(dataloader):
            if i == num_batch:
                break
            if verbose:
                print('Distillation: %s / %s' % (i + 1, num_batch))
            gaussian_data = gaussian_data.to(device)
            gaussian_data.requires_grad = True
            optimizer = optim.Adam([gaussian_data], lr=lr)
            scheduler = optim.lr_scheduler.ReduceLROnPlateau(optimizer, min_lr=1e-4, patience=25)
            for it in range(train_iter):
                teacher_model.zero_grad()
                optimizer.zero_grad()
                for hook in hooks:
                    hook.clear()
                length = torch.tensor([seqlen] * batch_size, device=device)
                encoded, encoded_len, encoded_sf = teacher_model(gaussian_data, length)
                teacher_model_decoder(encoder_output=encoded, encoder_output_scaling_factor=encoded_sf)
                total_loss = 0
                # statistics of every conv output against the running statistics of the BatchNorm behind it
                for (bn_mean, bn_std), hook in zip(bn_stats, hooks):
                    conv_output = hook.outputs
                    conv_mean = torch.mean(conv_output[0], dim=(0, 2))
                    conv_var = torch.var(conv_output[0] + eps, dim=(0, 2))
                    conv_std = torch.sqrt(conv_var + eps)
                    assert bn_mean.shape == conv_mean.shape and bn_std.shape == conv_var.shape
                    total_loss = total_loss + _kl_loss(bn_mean, bn_std, conv_mean, conv_std)
                total_loss.backward()
                optimizer.step()
                scheduler.step(total_loss.item())
                if history is not None:
                    history.append(float(total_loss.item()))
            refined_gaussian.append(gaussian_data.detach().clone())
    finally:
        torch.set_grad_enabled(was_enabled)
        for handle in hook_handles:
            handle.remove()
    return refined_gaussian
