"""EncDecCTCModel: preprocessor -> ConvASREncoder -> ConvASRDecoder -> greedy argmax
(nemo/collections/asr/models/ctc_models.py:43-147, 383-406), without the NeMo-core / Lightning layers.

Two execution paths share this one object:
  * host PyTorch (calibration with running ranges, dynamic mode, `--no_quant`): the modules' own forward;
  * the HIP engine: once every QuantAct is frozen (`qm.evaluate`), non-dynamic, symmetric and the BNs are
    folded, forward() packs the model once (qasr.pack) and runs mel front-end + encoder + decoder in the
    gfx950 kernels of libqasr_hip.so.  There is no CPU fallback for that configuration: a missing
    library or GPU raises.
"""
import io
import os
import tarfile

import numpy as np
import torch
import torch.nn as nn
import yaml

from nemo.collections.asr.modules.audio_preprocessing import AudioToMelSpectrogramPreprocessor
from nemo.collections.asr.modules.conv_asr import ConvASRDecoder, ConvASREncoder
from nemo.collections.asr.parts.jasper import MaskedConv1d
from nemo.quantization.utils.quant_modules import QuantAct
from qasr import configs as qconfigs
from qasr import synth, topology

_MODEL_CONFIG, _MODEL_WEIGHTS = "model_config.yaml", "model_weights.ckpt"


def _strip_target(d):
    return {k: v for k, v in d.items() if k not in ('_target_', 'cls', 'params')}


class EncDecCTCModel(nn.Module):
    def __init__(self, cfg, trainer=None):
        super().__init__()
        cfg = dict(cfg.get('model', cfg))
        enc, dec = dict(cfg['encoder']), dict(cfg['decoder'])
        # the fork hard-wires symmetric quantisation into both halves (ctc_models.py:103-107)
        enc['quant_mode'] = dec['quant_mode'] = 'symmetric'
        self.cfg = cfg
        self._cfg = cfg
        self.preprocessor = AudioToMelSpectrogramPreprocessor(**_strip_target(cfg['preprocessor']))
        self.encoder = ConvASREncoder(**_strip_target(enc))
        if dec.get('vocabulary') is None:
            dec['vocabulary'] = cfg.get('labels')
        self.decoder = ConvASRDecoder(**_strip_target(dec))
        self.spec_augmentation = None
        self._test_dl = None
        self._engine = None
        self._engine_key = None
        self._quant_version = 0

    # ------------------------------------------------------------------ construction / checkpoints
    @classmethod
    def list_available_models(cls):
        return ['QuartzNet15x5Base-En', 'Jasper10x5Dr-En']

    @classmethod
    def restore_from(cls, restore_path, map_location='cpu', strict=False):
        """.nemo = tar(.gz){model_config.yaml, model_weights.ckpt} (nemo/core/classes/modelPT.py:40-41,379-400)."""
        if not os.path.exists(restore_path):
            raise FileNotFoundError(f"Can't find {restore_path}")
        with tarfile.open(restore_path, 'r:*') as tar:
            names = {os.path.basename(m.name): m for m in tar.getmembers()}
            cfg = yaml.safe_load(tar.extractfile(names[_MODEL_CONFIG]).read())
            blob = tar.extractfile(names[_MODEL_WEIGHTS]).read()
        model = cls(cfg)
        sd = torch.load(io.BytesIO(blob), map_location=map_location, weights_only=True)
        model.load_state_dict(sd, strict=strict)
        return model

    def save_to(self, save_path):
        with tarfile.open(save_path, 'w:gz') as tar:
            def add(name, data):
                ti = tarfile.TarInfo(name)
                ti.size = len(data)
                tar.addfile(ti, io.BytesIO(data))
            add(_MODEL_CONFIG, yaml.safe_dump(self.cfg).encode())
            buf = io.BytesIO()
            torch.save(self.state_dict(), buf)
            add(_MODEL_WEIGHTS, buf.getvalue())

    @classmethod
    def from_pretrained(cls, model_name, refresh_cache=False):
        """The reference downloads `<name>.nemo` from NGC (ctc_models.py:55-88); there is no network here, so the
        file is looked up under $QASR_MODEL_DIR (or ~/.cache/torch/NeMo)."""
        if model_name not in cls.list_available_models():
            raise FileNotFoundError(f"Model {model_name} was not found. Available: {cls.list_available_models()}")
        for root in (os.environ.get('QASR_MODEL_DIR'), os.path.expanduser('~/.cache/torch/NeMo')):
            if root:
                for dirpath, _, files in os.walk(root):
                    if model_name + '.nemo' in files:
                        return cls.restore_from(os.path.join(dirpath, model_name + '.nemo'))
        raise FileNotFoundError(f"{model_name}.nemo not found locally and this environment has no network; put the "
                                f"checkpoint under $QASR_MODEL_DIR or pass a .nemo path")

    @classmethod
    def from_synthetic(cls, model_name='QuartzNet15x5Base-En', seed=0):
        """Random-init model of the named architecture with the deterministic weights of qasr.synth."""
        model = cls(qconfigs.model_config(model_name))
        sd = {k: torch.from_numpy(np.asarray(v)) for k, v in
              synth.make_state_dict(topology.MODELS[model_name](), seed).items()}
        missing, unexpected = model.load_state_dict(sd, strict=False)
        assert not unexpected, unexpected
        return model

    def load_state_dict(self, state_dict, strict=False):
        """strict=False by default like ModelPT (modelPT.py:400): the fork's extra buffers are absent from
        upstream checkpoints.  `...conv.weight` (QuantConv1d) is mirrored into the inner `...conv.conv.weight`."""
        sd = dict(state_dict)
        for k in list(sd):
            if k.endswith('.conv.weight') and k[:-len('weight')] + 'conv.weight' not in sd:
                sd[k[:-len('weight')] + 'conv.weight'] = sd[k]
        if 'decoder.decoder_layers.0.weight' in sd:
            sd.setdefault('decoder.decoder_layers.0.conv.weight', sd['decoder.decoder_layers.0.weight'])
            if 'decoder.decoder_layers.0.bias' in sd:
                sd.setdefault('decoder.decoder_layers.0.conv.bias', sd['decoder.decoder_layers.0.bias'])
        res = super().load_state_dict(sd, strict=strict)
        self._quant_state_changed()
        return res

    # ------------------------------------------------------------------ quantisation switches
    def set_quant_bit(self, quant_bit, mode='all'):
        self.encoder.set_quant_bit(quant_bit, mode)
        self.decoder.set_quant_bit(quant_bit, mode)
        self._quant_state_changed()

    def set_quant_mode(self, quant_mode):
        self.encoder.set_quant_mode(quant_mode)
        self.decoder.set_quant_mode(quant_mode)
        self._quant_state_changed()

    def _quant_state_changed(self):
        self._quant_version += 1
        if self._engine is not None:
            self._engine.close()
        self._engine = None

    # ------------------------------------------------------------------ data
    def setup_test_data(self, test_data_config):
        from nemo.collections.asr.data.audio_to_text import make_dataloader
        cfg = dict(test_data_config)
        cfg.setdefault('shuffle', False)
        self._test_dl = make_dataloader(cfg)

    def test_dataloader(self):
        return self._test_dl

    @torch.no_grad()
    def transcribe(self, paths2audio_files, batch_size=4, logprobs=False):
        """Greedy transcripts (or per-file log-probabilities) of audio files, in input order - the reference's debugging /
        prototyping entry (ctc_models.py:148-212, 476-503): dither off and pad_to 0 for the duration of the call, evaluation
        mode, a temporary manifest with `duration` 100000 and text 'nothing', batch size min(batch_size, #files), silence
        trimmed (`trim_silence: True`), everything restored afterwards.  A calibrated model runs on the HIP engine."""
        if paths2audio_files is None or len(paths2audio_files) == 0:
            return {}
        import json
        import tempfile
        from nemo.collections.asr.data.audio_to_text import make_dataloader
        from nemo.collections.asr.metrics.wer import WER
        hypotheses = []
        mode = self.training
        device = next(self.parameters()).device
        f = self.preprocessor.featurizer
        dither_value, pad_to_value = f.dither, f.pad_to
        try:
            f.dither, f.pad_to = 0.0, 0
            self.eval()
            with tempfile.TemporaryDirectory() as tmpdir:
                manifest = os.path.join(tmpdir, 'manifest.json')
                with open(manifest, 'w') as fp:
                    for audio_file in paths2audio_files:
                        fp.write(json.dumps({'audio_filepath': audio_file, 'duration': 100000, 'text': 'nothing'}) + '\n')
                loader = make_dataloader({'manifest_filepath': manifest, 'sample_rate': 16000, 'labels': self.decoder.vocabulary,
                                          'batch_size': min(batch_size, len(paths2audio_files)), 'trim_silence': True,
                                          'shuffle': False})
                wer = WER(vocabulary=self.decoder.vocabulary)
                for batch in loader:
                    logits, logits_len, greedy = self.forward(input_signal=batch[0].to(device).float(),
                                                              input_signal_length=batch[1].to(device))
                    if logprobs:
                        for idx in range(logits.shape[0]):
                            hypotheses.append(logits[idx][: logits_len[idx]])
                    else:
                        hypotheses += wer.ctc_decoder_predictions_tensor(greedy)
        finally:
            self.train(mode=mode)
            f.dither, f.pad_to = dither_value, pad_to_value
        return hypotheses

    # ------------------------------------------------------------------ engine path
    def _masked_convs(self):
        for blk in self.encoder.encoder_layers:
            yield from blk._masked_convs()

    def engine_ready(self):
        """True when the model is in the calibrated, frozen, folded, symmetric configuration the engine runs."""
        acts = [m for m in self.modules() if isinstance(m, QuantAct)]
        if not acts or any(a.quant_mode != 'symmetric' or a.running_stat or a.dynamic for a in acts):
            return False
        for blk in self.encoder.encoder_layers:
            if any(isinstance(l, nn.BatchNorm1d) for l in blk.mconv):
                return False                                    # BN not folded
        return all(mc.conv.fix_bn and mc.conv.quant_mode == 'symmetric' for mc in self._masked_convs())

    def dynamic_ready(self):
        """True when every QuantAct is in dynamic mode (qm.set_dynamic, quant_modules.py:149-167) on min / max or - all of
        them alike - percentile ranges (qm.set_percentile, :158-167) and the convs are folded and fixed: the configuration
        qasr.dynamic.DynamicRunner executes on the HIP kernels.  (Per-channel activation ranges are not a configuration the
        reference's QuantConv1d can run: int_conv multiplies [Cout,1,1] weight scales with [1,Cin,1] activation scales.)"""
        acts = [m for m in self.modules() if isinstance(m, QuantAct)]
        if not acts or any(a.quant_mode != 'symmetric' or not a.dynamic or a.per_channel for a in acts):
            return False
        if len({a.percentile or None for a in acts}) != 1:
            return False
        for blk in self.encoder.encoder_layers:
            if any(isinstance(l, nn.BatchNorm1d) for l in blk.mconv):
                return False
        return all(mc.conv.fix_bn and mc.conv.quant_mode == 'symmetric' for mc in self._masked_convs())

    def _get_dynamic_runner(self, device):
        """DynamicRunner for the live weights (both model families), or None if it declines the topology: such a model
        keeps the host modules in dynamic mode."""
        percentile = next(m for m in self.modules() if isinstance(m, QuantAct)).percentile or None
        key = ('dyn', self._quant_version, device.index or 0, percentile)
        if self._engine_key != key:
            from qasr import dynamic, engine as qengine
            qengine.load_library()
            cfg, sd, _, _, wbit, abit = self.export_pack_inputs()
            try:
                self._engine = dynamic.DynamicRunner(cfg, sd, wbit, abit, device, percentile=percentile)
            except NotImplementedError:
                self._engine = None
            self._engine_key = key
        return self._engine

    def export_pack_inputs(self):
        """(ModelCfg, pre-fold NeMo-keyed float state dict, act_min, act_max, wbit, abit) of the live model."""
        cfg = qconfigs.topology_from_config(self.cfg)
        plan = topology.conv_plan(cfg)
        sd, amin, amax = {}, [], []
        wbits, abits = set(), set()
        for blk, sites in zip(self.encoder.encoder_layers, plan):
            mcs = list(blk._masked_convs())
            assert len(mcs) == len(sites)
            for mc, s in zip(mcs, sites):
                sd[f'{s.key}.conv.weight'] = mc.conv.weight.detach().cpu().float()
                if mc.conv.bias is not None:
                    sd[f'{s.key}.conv.bias'] = mc.conv.bias.detach().cpu().float()
                if s.bn_key is not None:
                    bn = mc.conv.bn
                    assert bn is not None, f'{s.key}: call encoder.bn_folding() first'
                    for n in ('weight', 'bias', 'running_mean', 'running_var'):
                        sd[f'{s.bn_key}.{n}'] = getattr(bn, n).detach().cpu().float()
                amin.append(float(mc.act.x_min))
                amax.append(float(mc.act.x_max))
                wbits.add(mc.conv.weight_bit)
                abits.add(mc.act.activation_bit - (1 if mc.asymmetric else 0))
            amin.append(float(blk.res_act.x_min))
            amax.append(float(blk.res_act.x_max))
        q = self.decoder.decoder_layers[0]
        sd['decoder.decoder_layers.0.weight'] = q.weight.detach().cpu().float()
        sd['decoder.decoder_layers.0.bias'] = q.bias.detach().cpu().float()
        amin.append(float(self.decoder.act.x_min))
        amax.append(float(self.decoder.act.x_max))
        wbits.add(q.weight_bit)
        abits.add(self.decoder.act.activation_bit)
        if len(wbits) != 1 or len(abits) != 1:
            raise NotImplementedError(f'mixed bit-widths are not packed yet: weights {wbits}, activations {abits}')
        return cfg, sd, np.array(amin, np.float32), np.array(amax, np.float32), wbits.pop(), abits.pop()

    def _get_engine(self, device):
        key = (self._quant_version, device.index or 0)
        if self._engine is None or self._engine_key != key:
            from qasr import engine as qengine, pack
            qengine.load_library()                               # raises when the HIP extension is missing
            blob, self._pack_meta = pack.pack_model(*self.export_pack_inputs())
            self._engine = qengine.Engine(blob, device.index or 0)
            self._engine_key = key
        return self._engine

    def _frontend_hip_supported(self):
        """The HIP front-end kernels are built for the QuartzNet / Jasper preprocessor (quartznet_15x5.yaml:30-41):
        n_fft 512, hop 160, a 320-tap window, per-feature normalisation, log(x + 2^-24), power spectrum.  Any other
        featurizer configuration runs the host module on the GPU tensors instead - the same module calibration used -
        rather than silently producing different features (a shorter window would even be read out of bounds)."""
        f = self.preprocessor.featurizer
        return (f.n_fft == 512 and f.hop_length == 160 and f.win_length == 320 and f.window.numel() == 320
                and f.fb.dim() == 3 and f.fb.shape[2] == 257 and f.normalize == 'per_feature' and bool(f.log)
                and f.log_zero_guard_type == 'add' and not isinstance(f.log_zero_guard_value, str)
                and float(f.log_zero_guard_value) == 2.0 ** -24 and float(f.mag_power) == 2.0 and f.preemph is not None
                and f.pad_value == 0 and isinstance(f.pad_to, int) and f.pad_to >= 0)

    def _frontend_plan_for(self, device):
        """(filterbank on `device`, qasr_frontend_plan workspace): the filterbank-only tables, built once per model / device."""
        from qasr import engine as qengine
        f = self.preprocessor.featurizer
        fb = f.fb[0].to(device=device, dtype=torch.float32).contiguous()
        key = (fb.data_ptr(), f.fb._version, str(device))
        if getattr(self, '_frontend_plan_key', None) != key:
            self._frontend_plan = qengine.frontend_plan(fb)
            self._frontend_plan_key = key
        return self._frontend_plan._qasr_fb, self._frontend_plan

    def _frontend_hip(self, signal, length):
        from qasr import engine as qengine
        f = self.preprocessor.featurizer
        if not self._frontend_hip_supported():
            return self.preprocessor(input_signal=signal, length=length)
        if f.dither > 0:
            signal = signal + f.dither * torch.randn_like(signal)
        fb, plan = self._frontend_plan_for(signal.device)
        return qengine.frontend_mel(signal.float().contiguous(), length, fb, f.window.contiguous(), float(f.preemph),
                                    int(f.pad_to), plan=plan)

    # ------------------------------------------------------------------ forward
    def forward(self, input_signal=None, input_signal_length=None, processed_signal=None,
                processed_signal_length=None):
        has_in = input_signal is not None and input_signal_length is not None
        has_pr = processed_signal is not None and processed_signal_length is not None
        if has_in == has_pr:
            raise ValueError(f"{self} Arguments ``input_signal`` and ``input_signal_length`` are mutually exclusive "
                             " with ``processed_signal`` and ``processed_signal_len`` arguments.")
        ref = input_signal if has_in else processed_signal
        if self.engine_ready():
            if not ref.is_cuda:
                raise RuntimeError('the calibrated integer model runs on the MI355X HIP engine only: move the inputs '
                                   'to cuda (there is no CPU fallback for the quantised inference path)')
            eng = self._get_engine(ref.device)
            f = self.preprocessor.featurizer
            if has_in and self._frontend_hip_supported() and f.pad_to > 0:
                # audio -> tokens as one engine call (qasr_engine_forward_audio): front-end, encoder and decoder replay as
                # one hipGraph launch when the caller keeps its buffers
                sig = input_signal.float().contiguous()
                if f.dither > 0:
                    sig = sig + f.dither * torch.randn_like(sig)
                fb, plan = self._frontend_plan_for(ref.device)
                log_probs, tokens, enc_len = eng.forward_audio(
                    sig, input_signal_length.to(device=ref.device, dtype=torch.int32).contiguous(), fb,
                    f.window.to(device=ref.device, dtype=torch.float32).contiguous(), plan, float(f.preemph), int(f.pad_to))
                return log_probs, enc_len.long(), tokens.long()
            if has_in:
                processed_signal, processed_signal_length = self._frontend_hip(input_signal, input_signal_length)
            log_probs, tokens, enc_len = eng.forward(processed_signal.float(), processed_signal_length)
            return log_probs, enc_len.long(), tokens.long()
        if ref.is_cuda and self.dynamic_ready():
            runner = self._get_dynamic_runner(ref.device)
            if runner is not None:                               # dynamic-quantisation device path
                if has_in:
                    processed_signal, processed_signal_length = self._frontend_hip(input_signal, input_signal_length)
                out = runner.forward(processed_signal.float(), processed_signal_length)
                return out['log_probs'], out['enc_len'].long(), out['tokens'].long()
        if has_in:
            processed_signal, processed_signal_length = self.preprocessor(input_signal=input_signal,
                                                                          length=input_signal_length)
        encoded, encoded_len, encoded_sf = self.encoder(audio_signal=processed_signal, length=processed_signal_length)
        log_probs = self.decoder(encoder_output=encoded, encoder_output_scaling_factor=encoded_sf)
        return log_probs, encoded_len, log_probs.argmax(dim=-1, keepdim=False)
