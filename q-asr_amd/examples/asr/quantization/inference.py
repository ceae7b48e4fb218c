#!/usr/bin/env python3
"""Calibrate + evaluate a quantised CTC model — the reference's entry point
(examples/asr/quantization/inference.py:46-159) with the same flags, running on one MI355X.

    python inference.py --asr_model QuartzNet15x5Base-En.nemo --dataset dev_clean.json \
        --load synthetic.pkl --weight_bit 8 --act_bit 8 --percentile 99.996 --batch_size 32

Flow (identical to the reference): load model -> set bit-widths -> percentile -> BN fold -> calibrate the
QuantAct ranges on the synthetic (mel-domain) batches in host PyTorch -> `qm.evaluate` -> evaluation loop, which
now runs mel front-end + integer encoder/decoder in the HIP engine -> greedy CTC decode -> WER.
`--load` reads what the reference's synthesize.py writes (`pickle.dump([x.cpu() ...])`, synthesize.py:103-104) through
a restricted unpickler that can only rebuild tensors, or a .pt / .npz written by this repo; `--synthetic_calib N`
generates N seeded batches instead.
"""
import os
import sys
import time
from argparse import ArgumentParser

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(_HERE, '..', '..', '..')))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import nemo.quantization.utils.quantize_model as qm  # noqa: E402
from nemo.collections.asr.metrics.wer import WER, word_error_rate  # noqa: E402
from nemo.collections.asr.models import EncDecCTCModel  # noqa: E402

if not torch.cuda.is_available():
    raise Exception("Current implementation only supports GPU (MI355X / ROCm)")


from qasr.calib_io import load_synthetic  # noqa: E402  (restricted loader for --load)


def main():
    p = ArgumentParser()
    p.add_argument("--asr_model", type=str, default="QuartzNet15x5Base-En", required=True)
    p.add_argument("--dataset", type=str, required=True, help="path to evaluation data (JSON-lines manifest)")
    p.add_argument("--batch_size", type=int, default=8)
    p.add_argument("--normalize_text", default=True, type=bool)
    p.add_argument("--shuffle", action='store_true')
    p.add_argument("--load", type=str, default=None, help="load path for the synthetic data")
    p.add_argument("--percentile", type=float, default=None)
    p.add_argument("--weight_bit", type=int, default=8)
    p.add_argument("--act_bit", type=int, default=8)
    p.add_argument("--dynamic", action='store_true')
    p.add_argument("--no_quant", action='store_true')
    p.add_argument("--eval_early_stop", type=int, default=None)
    p.add_argument("--calib_early_stop", type=int, default=None)
    p.add_argument("--synthetic_calib", type=int, default=0, help="(extension) generate N seeded calibration batches")
    p.add_argument("--synthetic_model", action='store_true', help="(extension) random-init weights of --asr_model")
    p.add_argument("--dither", type=float, default=None, help="(extension) override the preprocessor's dither (0: reproducible runs)")
    p.add_argument("--dump_hyps", type=str, default=None, help="(extension) write hypotheses, references and WER as JSON")
    args = p.parse_args()
    torch.set_grad_enabled(False)

    if args.asr_model.endswith('.nemo'):
        asr_model = EncDecCTCModel.restore_from(restore_path=args.asr_model)
    elif args.synthetic_model:
        asr_model = EncDecCTCModel.from_synthetic(args.asr_model)
    else:
        asr_model = EncDecCTCModel.from_pretrained(model_name=args.asr_model)
    asr_model = asr_model.cuda()
    if args.dither is not None:
        asr_model.preprocessor.featurizer.dither = args.dither
    asr_model.setup_test_data(test_data_config={
        'sample_rate': 16000, 'manifest_filepath': args.dataset, 'labels': asr_model.decoder.vocabulary,
        'batch_size': args.batch_size, 'normalize_transcripts': args.normalize_text, 'shuffle': args.shuffle})

    distilled = None
    if args.load is not None:
        print('Data loaded from %s' % args.load)
        distilled = load_synthetic(args.load)
    elif args.synthetic_calib:
        from qasr import synth
        distilled = [torch.from_numpy(a) for a in synth.make_calibration(args.synthetic_calib, args.batch_size, 64, 500)]
    else:
        assert args.dynamic, "synthetic data must be loaded unless running with the dynamic quantization mode"

    asr_model.eval()
    asr_model.set_quant_bit(args.weight_bit, mode='weight')
    asr_model.set_quant_bit(args.act_bit, mode='act')
    if args.percentile is not None:
        qm.set_percentile(asr_model, args.percentile)
    if args.no_quant:
        asr_model.set_quant_mode('none')
    else:
        asr_model.encoder.bn_folding()

    if not args.dynamic and not args.no_quant:
        print('Calibrating...')
        qm.calibrate(asr_model)
        bs, _, seqlen = distilled[0].shape
        length = torch.tensor([seqlen] * bs).cuda()
        for i, inputs in enumerate(distilled):
            if args.calib_early_stop is not None and i == args.calib_early_stop:
                break
            enc, enc_len, enc_sf = asr_model.encoder(audio_signal=inputs.cuda(), length=length)
            asr_model.decoder(encoder_output=enc, encoder_output_scaling_factor=enc_sf)

    print('Evaluating...')
    qm.evaluate(asr_model)
    qm.set_dynamic(asr_model, args.dynamic)
    labels_map = dict(enumerate(asr_model.decoder.vocabulary))
    wer = WER(vocabulary=asr_model.decoder.vocabulary)
    hyps, refs = [], []
    audio_s, t0 = 0.0, time.time()
    for i, batch in enumerate(asr_model.test_dataloader()):
        if i == args.eval_early_stop:
            break
        batch = [x.cuda() for x in batch]
        log_probs, enc_len, greedy = asr_model(input_signal=batch[0].float(), input_signal_length=batch[1])
        hyps += wer.ctc_decoder_predictions_tensor(greedy)
        for row in batch[2].cpu().numpy():
            refs.append(''.join(labels_map[c] for c in row))
        audio_s += float(batch[1].sum()) / 16000.0
    torch.cuda.synchronize()
    wall = time.time() - t0
    served = type(getattr(asr_model, '_engine', None)).__name__
    print('path:', {'Engine': 'static integer engine (HIP)', 'DynamicRunner': 'dynamic device path (HIP)'}.get(
        served, 'host modules'))
    wer_value = word_error_rate(hypotheses=hyps, references=refs)
    print('WER:', wer_value)
    if args.dump_hyps:
        import json
        with open(args.dump_hyps, 'w') as f:
            json.dump(dict(hypotheses=hyps, references=refs, wer=wer_value, path=served), f)
    print(f'RTFx (incl. host data loading): {audio_s / max(wall, 1e-9):.1f}  ({audio_s:.1f} s audio in {wall:.2f} s)')


if __name__ == '__main__':
    main()
