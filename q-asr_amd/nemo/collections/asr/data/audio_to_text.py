"""Minimal evaluation data path: JSON-lines manifest -> 16-bit PCM WAV -> padded batch.
Mirrors the behaviour inference.py relies on: AudioToCharDataset items (audio, audio_len, tokens, tokens_len),
pad-collate with pad id 0 (nemo/collections/asr/data/audio_to_text.py:41-78,81-291), LibriSpeech manifest
fields {audio_filepath, duration, text} (scripts/get_librispeech_data.py:105-120), transcripts through
parsers.make_parser(labels, 'en', unk_id=-1, blank_id=-1, do_normalize=...) as AudioToCharDataset builds it
(audio_to_text.py:258-267: ENCharParser = cleaners.clean_text + per-character lookup)."""
import json
import re
import wave

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset


def read_wav(path, target_sr=16000):
    """16-bit PCM WAV -> float32 in [-1, 1) (segment.py:95-133 divides int samples by 2^(bits-1))."""
    with wave.open(path, 'rb') as w:
        sr, ch, width, n = w.getframerate(), w.getnchannels(), w.getsampwidth(), w.getnframes()
        raw = w.readframes(n)
    if width != 2:
        raise ValueError(f'{path}: only 16-bit PCM is supported (got {8 * width}-bit)')
    x = np.frombuffer(raw, dtype='<i2').astype(np.float32) / 32768.0
    if ch > 1:
        x = x.reshape(-1, ch).mean(axis=1)
    if sr != target_sr:
        raise ValueError(f'{path}: sample rate {sr} != {target_sr} (resampling is not part of the hot path)')
    return x


def trim_silence(x, top_db=60.0, frame_length=2048, hop_length=512):
    """Leading / trailing silence removed the way AudioSegment does with trim=True (segment.py:60-61:
    `librosa.effects.trim(samples, 60)`): frame-wise RMS over centred frames (reflect padding) in dB relative to the loudest
    frame; everything before the first / after the last frame above -top_db goes.  librosa is not importable here: restated
    from its documented behaviour, parity unpinned."""
    x = np.asarray(x, dtype=np.float32)
    if x.size == 0:
        return x
    pad = frame_length // 2
    xp = np.pad(x, pad, mode='reflect') if x.size > pad else np.pad(x, pad, mode='constant')
    n_frames = 1 + (xp.size - frame_length) // hop_length
    idx = np.arange(frame_length)[None, :] + hop_length * np.arange(n_frames)[:, None]
    rms = np.sqrt(np.mean(xp[idx].astype(np.float64) ** 2, axis=1))
    ref = rms.max()
    if ref <= 0:
        return x[:0]
    db = 20.0 * np.log10(np.maximum(rms, 1e-10) / ref)       # amplitude_to_db(rms, ref=np.max, top_db=None)
    keep = np.nonzero(db > -top_db)[0]
    start = int(keep[0]) * hop_length
    end = min(x.size, (int(keep[-1]) + 1) * hop_length)
    return x[start:end]


def normalize_text(text, vocabulary):
    """Normalised transcript as the reference's parser sees it (ENCharParser._normalize, parsers.py:136-145)."""
    from nemo.collections.asr.parts import parsers
    out = parsers.make_parser(labels=list(vocabulary), name='en', unk_id=-1, blank_id=-1, do_normalize=True)._normalize(text)
    return '' if out is None else out


class AudioToCharDataset(Dataset):
    def __init__(self, manifest_filepath, labels, sample_rate=16000, normalize=True, max_duration=None,
                 min_duration=None, trim=False, **_unused):
        self.labels = list(labels)
        self.index = {c: i for i, c in enumerate(self.labels)}
        self.sample_rate = sample_rate
        self.trim = bool(trim)                               # `trim_silence` of the reference's dataset config (audio_to_text.py:228)
        self.items = []
        for path in str(manifest_filepath).split(','):
            with open(path) as f:
                for line in f:
                    if not line.strip():
                        continue
                    it = json.loads(line)
                    dur = it.get('duration')
                    if dur is not None and ((max_duration and dur > max_duration) or (min_duration and dur < min_duration)):
                        continue
                    text = it.get('text', '')
                    if normalize:
                        text = normalize_text(text, self.labels)
                    self.items.append((it['audio_filepath'], text))

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        path, text = self.items[i]
        x = read_wav(path, self.sample_rate)
        if self.trim:
            x = trim_silence(x)
        x = torch.from_numpy(np.ascontiguousarray(x))
        t = torch.tensor([self.index[c] for c in text if c in self.index], dtype=torch.long)
        return x, torch.tensor(x.numel(), dtype=torch.long), t, torch.tensor(t.numel(), dtype=torch.long)

    @staticmethod
    def collate_fn(batch, pad_id=0):
        al = max(int(b[1]) for b in batch)
        tl = max(int(b[3]) for b in batch) if batch else 0
        audio = torch.zeros(len(batch), al)
        toks = torch.full((len(batch), tl), pad_id, dtype=torch.long)
        for i, (x, n, t, m) in enumerate(batch):
            audio[i, :int(n)] = x
            toks[i, :int(m)] = t
        return audio, torch.stack([b[1] for b in batch]), toks, torch.stack([b[3] for b in batch])


def make_dataloader(config):
    ds = AudioToCharDataset(config['manifest_filepath'], config['labels'], sample_rate=config.get('sample_rate', 16000),
                            normalize=config.get('normalize_transcripts', True),
                            max_duration=config.get('max_duration'), min_duration=config.get('min_duration'),
                            trim=config.get('trim_silence', False))
    return DataLoader(ds, batch_size=config['batch_size'], shuffle=config.get('shuffle', False),
                      collate_fn=AudioToCharDataset.collate_fn, drop_last=config.get('drop_last', False),
                      num_workers=config.get('num_workers', 0))
