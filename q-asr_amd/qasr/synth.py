"""Deterministic synthetic checkpoints and inputs (there is no network for real ones).

``make_state_dict`` emits a dict of float32 numpy arrays keyed exactly like an
upstream-NeMo ``model_weights.ckpt`` for the encoder/decoder (pre-fold
module indices, ``...mconv.{i}.conv.weight``, BatchNorm at ``...mconv.{i+1|i+2}.*``,
``...res.{j}.{0,1}``, ``decoder.decoder_layers.0.{weight,bias}``).  The generator is
numpy's PCG64 so the same seed gives the same bytes on any machine / torch version;
the golden-vector script feeds these arrays to the reference modules and the tests
feed them to this repo's own loader.

BN statistics follow the survey's recipe (mean N(0,.1), var U(.3,.7), gamma
U(.5,1.5), beta N(0,.3)); plain ``init_weights`` BN (jasper.py:43-50) lets the
activations decay to ~1e-13 after a few blocks.
"""
import numpy as np

from .topology import ModelCfg, conv_plan


def _xavier(rng, shape):
    cout, cin_g, k = shape
    fan_in, fan_out = cin_g * k, cout * k
    a = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-a, a, size=shape).astype(np.float32)


def make_state_dict(cfg: ModelCfg, seed: int = 0, gain: float = 1.0):
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = {}
    for sites in conv_plan(cfg):
        for s in sites:
            w = _xavier(rng, (s.cout, s.cin // s.groups, s.kernel))
            if s.role == 'dw':
                w = w * np.float32(2.0 * gain)   # depthwise taps: keep the signal alive through 15 blocks
            sd[f'{s.key}.conv.weight'] = w.astype(np.float32)
            if s.bn_key is not None:
                c = s.cout
                sd[f'{s.bn_key}.weight'] = rng.uniform(0.5, 1.5, c).astype(np.float32)
                sd[f'{s.bn_key}.bias'] = rng.normal(0.0, 0.3, c).astype(np.float32)
                sd[f'{s.bn_key}.running_mean'] = rng.normal(0.0, 0.1, c).astype(np.float32)
                sd[f'{s.bn_key}.running_var'] = rng.uniform(0.3, 0.7, c).astype(np.float32)
                sd[f'{s.bn_key}.num_batches_tracked'] = np.array(1, dtype=np.int64)
    cdec = cfg.blocks[-1].filters
    ncls = cfg.num_classes + 1
    # x3 / small bias: keeps the greedy tokens input-dependent instead of bias-dominated
    sd['decoder.decoder_layers.0.weight'] = (_xavier(rng, (ncls, cdec, 1)) * np.float32(3.0)).astype(np.float32)
    sd['decoder.decoder_layers.0.bias'] = rng.normal(0.0, 0.1, ncls).astype(np.float32)
    return sd


def make_features(batch: int, feat: int, frames: int, seed: int = 0):
    """Normalised-mel-like encoder input ``[B, feat, frames]`` ~ N(0,1), float32."""
    rng = np.random.Generator(np.random.PCG64(1000 + seed))
    return rng.standard_normal((batch, feat, frames)).astype(np.float32)


def make_calibration(nbatches: int, batch: int, feat: int, frames: int, seed: int = 0):
    """Calibration batches in the spirit of the reference's zero-shot data: the
    distilled tensors start from U(-0.3,0.3)+noise and converge towards normalised
    mel statistics (distill_data.py:11-25, 143-152); we use N(0,1) clipped to +-4."""
    rng = np.random.Generator(np.random.PCG64(2000 + seed))
    return [np.clip(rng.standard_normal((batch, feat, frames)), -4, 4).astype(np.float32)
            for _ in range(nbatches)]


def make_audio(batch: int, samples: int, seed: int = 0, amp: float = 0.1):
    """Synthetic 16 kHz waveform batch ``[B, samples]`` = amp * N(0,1)."""
    rng = np.random.Generator(np.random.PCG64(3000 + seed))
    return (amp * rng.standard_normal((batch, samples))).astype(np.float32)
