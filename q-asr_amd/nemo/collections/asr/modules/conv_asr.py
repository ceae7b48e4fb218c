"""ConvASREncoder / ConvASRDecoder (nemo/collections/asr/modules/conv_asr.py:47-221, 223-319) for the
QuartzNet / Jasper CTC path.  Same constructor arguments, forward contract and state-dict keys;
the NeuralModule / typecheck / export machinery of NeMo core is not part of the hot path."""
from typing import Optional

import torch
import torch.nn as nn

from nemo.collections.asr.parts.jasper import JasperBlock, init_weights, jasper_activations
from nemo.quantization.utils.quant_modules import QuantAct, QuantConv1d


def _as_dict(cfg):
    return dict(cfg) if not isinstance(cfg, dict) else cfg


class ConvASREncoder(nn.Module):
    def __init__(self, jasper, activation: str, feat_in: int, normalization_mode: str = "batch",
                 residual_mode: str = "add", norm_groups: int = -1, conv_mask: bool = True, frame_splicing: int = 1,
                 init_mode: Optional[str] = 'xavier_uniform', quant_mode: Optional[str] = 'none',
                 quant_bit: Optional[int] = 8):
        super().__init__()
        act = jasper_activations[activation]()
        feat_in = feat_in * frame_splicing
        self._feat_in = feat_in
        self.quant_mode = quant_mode
        self.convs_before_bn = []
        panes, layers = [], []
        self.dense_residual = False
        for i, lcfg in enumerate(jasper):
            lcfg = _as_dict(lcfg)
            dense = []
            if lcfg.get('residual_dense', False):
                panes.append(feat_in)
                dense = panes
                self.dense_residual = True
            blk = JasperBlock(
                feat_in, lcfg['filters'], repeat=lcfg['repeat'], kernel_size=lcfg['kernel'], stride=lcfg['stride'],
                dilation=lcfg['dilation'], dropout=lcfg.get('dropout', 0.0), residual=lcfg['residual'],
                groups=lcfg.get('groups', 1), separable=lcfg.get('separable', False), heads=lcfg.get('heads', -1),
                residual_mode=lcfg.get('residual_mode', residual_mode), normalization=normalization_mode,
                norm_groups=norm_groups, activation=act, residual_panes=dense, conv_mask=conv_mask,
                se=lcfg.get('se', False), kernel_size_factor=lcfg.get('kernel_size_factor', 1.0),
                stride_last=lcfg.get('stride_last', False), quant_mode=quant_mode, quant_bit=quant_bit, layer_num=i)
            layers.append(blk)
            self.convs_before_bn += blk.convs_before_bn
            feat_in = lcfg['filters']
        self._feat_out = feat_in
        self.encoder_layers = layers
        self.encoder = nn.Sequential(*layers)
        self.apply(lambda m: init_weights(m, mode=init_mode))

    def forward(self, audio_signal, length=None, audio_signal_scaling_factor=None):
        s_input = [(audio_signal, audio_signal_scaling_factor)]
        for layer in self.encoder_layers:
            s_input, length = layer((s_input, length))
        out, sf = s_input[-1]
        if length is None:
            assert self.quant_mode != 'symmetric'
            return out
        return out, length, sf

    def bn_folding(self):
        for l in self.encoder_layers:
            l.bn_folding()

    def set_quant_bit(self, quant_bit, mode='all'):
        assert mode in ['all', 'weight', 'act']
        for l in self.encoder_layers:
            l.set_quant_bit(quant_bit, mode)

    def set_quant_mode(self, quant_mode):
        self.quant_mode = quant_mode
        for l in self.encoder_layers:
            l.set_quant_mode(quant_mode)


class ConvASRDecoder(nn.Module):
    """QuantAct -> 1x1 QuantConv1d(feat_in -> num_classes+1, real bias, no BN) -> log_softmax."""

    def __init__(self, feat_in, num_classes, init_mode="xavier_uniform", vocabulary=None, quant_mode='none',
                 quant_bit=8):
        super().__init__()
        self.quant_mode = quant_mode
        if vocabulary is not None and num_classes != len(vocabulary):
            raise ValueError(f"If vocabulary is specified, it's length should be equal to the num_classes. "
                             f"Instead got: num_classes={num_classes} and len(vocabulary)={len(vocabulary)}")
        self.__vocabulary = None if vocabulary is None else list(vocabulary)
        self._feat_in = feat_in
        self._num_classes = num_classes + 1                  # + CTC blank
        self.act = QuantAct(quant_bit, quant_mode=quant_mode, per_channel=False)
        qconv = QuantConv1d(quant_bit, bias_bit=32, quant_mode=quant_mode, per_channel=True)
        qconv.set_param(nn.Conv1d(feat_in, self._num_classes, kernel_size=1, bias=True))
        self.decoder_layers = nn.Sequential(qconv)
        self.apply(lambda m: init_weights(m, mode=init_mode))

    def forward(self, encoder_output, encoder_output_scaling_factor=None):
        out, sf = self.act(encoder_output, encoder_output_scaling_factor)
        for l in self.decoder_layers:
            out, _ = l(out, sf)
        return torch.nn.functional.log_softmax(out.transpose(1, 2), dim=-1)

    def set_quant_bit(self, quant_bit, mode='all'):
        assert mode in ['all', 'weight', 'act']
        if mode in ('all', 'act'):
            self.act.activation_bit = quant_bit
        if mode in ('all', 'weight'):
            for l in self.decoder_layers:
                l.weight_bit = quant_bit

    def set_quant_mode(self, quant_mode):
        self.quant_mode = self.act.quant_mode = quant_mode
        for l in self.decoder_layers:
            l.quant_mode = quant_mode

    @property
    def vocabulary(self):
        return self.__vocabulary

    @property
    def num_classes_with_blank(self):
        return self._num_classes
