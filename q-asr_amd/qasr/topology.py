"""Encoder topologies for the two model families on the hot path.

The block tables restate the hyper-parameters of the reference's model configs
(examples/asr/conf/quartznet_15x5.yaml:55-215, jasper_10x5dr.yaml:51-152) as
plain Python data, and ``conv_plan`` expands them into the flat list of
MaskedConv1d sites exactly as ConvASREncoder.__init__ / JasperBlock.__init__ do
(nemo/collections/asr/modules/conv_asr.py:136-192,
nemo/collections/asr/parts/jasper.py:349-447,548-633).
"""
from dataclasses import dataclass, field
from typing import List, Optional

VOCABULARY = [' ', 'a', 'b', 'c', 'd', 'e', 'f', 'g', 'h', 'i', 'j', 'k', 'l', 'm', 'n', 'o',
              'p', 'q', 'r', 's', 't', 'u', 'v', 'w', 'x', 'y', 'z', "'"]


@dataclass
class BlockCfg:
    filters: int
    kernel: int
    repeat: int
    stride: int = 1
    dilation: int = 1
    residual: bool = False
    separable: bool = False
    residual_dense: bool = False


@dataclass
class ModelCfg:
    name: str
    feat_in: int
    blocks: List[BlockCfg]
    num_classes: int = 28  # + 1 blank inside the decoder
    vocabulary: List[str] = field(default_factory=lambda: list(VOCABULARY))


def quartznet15x5() -> ModelCfg:
    b = [BlockCfg(256, 33, 1, stride=2, separable=True)]
    for k, f in ((33, 256), (39, 256), (51, 512), (63, 512), (75, 512)):
        b += [BlockCfg(f, k, 5, residual=True, separable=True) for _ in range(3)]
    b.append(BlockCfg(512, 87, 1, dilation=2, separable=True))
    b.append(BlockCfg(1024, 1, 1))
    return ModelCfg('QuartzNet15x5Base-En', 64, b)


def jasper10x5dr() -> ModelCfg:
    b = [BlockCfg(256, 11, 1, stride=2)]
    for k, f in ((11, 256), (13, 384), (17, 512), (21, 640), (25, 768)):
        b += [BlockCfg(f, k, 5, residual=True, residual_dense=True) for _ in range(2)]
    b.append(BlockCfg(896, 29, 1, dilation=2))
    b.append(BlockCfg(1024, 1, 1))
    return ModelCfg('Jasper10x5Dr-En', 64, b)


def mini_quartznet(c0=32, c1=48) -> ModelCfg:
    """Small QuartzNet-shaped net (every block kind once) for parity tests."""
    b = [BlockCfg(c0, 11, 1, stride=2, separable=True),
         BlockCfg(c0, 11, 3, residual=True, separable=True),
         BlockCfg(c1, 13, 2, residual=True, separable=True),
         BlockCfg(c1, 15, 1, dilation=2, separable=True),
         BlockCfg(64, 1, 1)]
    return ModelCfg('MiniQuartzNet', 16, b)


def mini_jasper() -> ModelCfg:
    """Small Jasper-shaped net: dense convs + dense residual."""
    b = [BlockCfg(32, 5, 1, stride=2),
         BlockCfg(32, 5, 2, residual=True, residual_dense=True),
         BlockCfg(48, 7, 2, residual=True, residual_dense=True),
         BlockCfg(48, 9, 1, dilation=2),
         BlockCfg(64, 1, 1)]
    return ModelCfg('MiniJasper', 16, b)


MODELS = {
    'QuartzNet15x5Base-En': quartznet15x5,
    'Jasper10x5Dr-En': jasper10x5dr,
    'MiniQuartzNet': mini_quartznet,
    'MiniJasper': mini_jasper,
}


def same_padding(kernel: int, stride: int, dilation: int) -> int:
    """jasper.py:61-66."""
    if stride > 1 and dilation > 1:
        raise ValueError("Only stride OR dilation may be greater than 1")
    if dilation > 1:
        return (dilation * kernel) // 2 - 1
    return kernel // 2


@dataclass
class ConvSite:
    """One MaskedConv1d (QuantAct + QuantConv1d) of the encoder."""
    key: str            # state-dict prefix, e.g. 'encoder.encoder.3.mconv.5'
    bn_key: Optional[str]   # state-dict prefix of the BatchNorm folded into it
    block: int
    role: str           # 'dw' | 'pw' | 'dense' | 'res'
    cin: int
    cout: int
    kernel: int
    stride: int
    dilation: int
    padding: int
    groups: int
    asymmetric: bool    # act bits = quant_bit + 1 (jasper.py:159-163)
    relu_after: bool    # followed by ReLU inside the block's mconv list
    pane: int = -1      # residual pane index (role == 'res')


def conv_plan(cfg: ModelCfg) -> List[List[ConvSite]]:
    """Per block: the MaskedConv1d sites in forward order (mconv first, then res).

    Pre-fold module indices (what a .nemo checkpoint uses): a separable repeat is
    [dw, pw, BN, ReLU, Dropout] (5 slots), a dense repeat [conv, BN, ReLU, Dropout]
    (4 slots); the last repeat has no ReLU/Dropout in mconv (jasper.py:349-396).
    """
    plan = []
    feat_in = cfg.feat_in
    panes: List[int] = []
    for bi, b in enumerate(cfg.blocks):
        sites = []
        pad = same_padding(b.kernel, b.stride, b.dilation)
        if b.residual_dense:
            panes.append(feat_in)
        cin = feat_in
        idx = 0
        for r in range(b.repeat):
            last = r == b.repeat - 1
            first_layer = bi == 0 and r == 0
            pre = f'encoder.encoder.{bi}.mconv'
            if b.separable:
                sites.append(ConvSite(f'{pre}.{idx}', None, bi, 'dw', cin, cin, b.kernel, b.stride,
                                      b.dilation, pad, cin, not first_layer, False))
                sites.append(ConvSite(f'{pre}.{idx + 1}', f'{pre}.{idx + 2}', bi, 'pw', cin, b.filters,
                                      1, 1, 1, 0, 1, False, not last))
                idx += 5
            else:
                sites.append(ConvSite(f'{pre}.{idx}', f'{pre}.{idx + 1}', bi, 'dense', cin, b.filters,
                                      b.kernel, b.stride, b.dilation, pad, 1, not first_layer, not last))
                idx += 4
            cin = b.filters
        if b.residual:
            res_panes = list(panes) if b.residual_dense else [feat_in]
            for j, ip in enumerate(res_panes):
                pre = f'encoder.encoder.{bi}.res.{j}'
                sites.append(ConvSite(f'{pre}.0', f'{pre}.1', bi, 'res', ip, b.filters, 1, 1, 1, 0, 1,
                                      bi != 0, False, pane=j))
        plan.append(sites)
        feat_in = b.filters
    return plan


def out_len(length: int, s: ConvSite) -> int:
    """MaskedConv1d.get_seq_len (jasper.py:170-173)."""
    return (length + 2 * s.padding - s.dilation * (s.kernel - 1) - 1) // s.stride + 1
