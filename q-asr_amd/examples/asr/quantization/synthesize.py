#!/usr/bin/env python3
"""Synthesize zero-shot calibration data from the BatchNorm statistics of a float model - the reference's
examples/asr/quantization/synthesize.py:48-104 with the same flags.  Output: `<prefix>_nb<N>_iter<I>_lr<lr>.pkl`, a
pickled list of CPU tensors [batch, 64, seqlen], exactly what `inference.py --load` of either code base reads.

    python synthesize.py --asr_model QuartzNet15x5Base-En.nemo --dataset dev_clean.json --num_batch 50 --dump_path out/
"""
import os
import pickle
import sys
from argparse import ArgumentParser

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(_HERE, '..', '..', '..')))

import torch  # noqa: E402

from nemo.collections.asr.models import EncDecCTCModel  # noqa: E402
from nemo.quantization.utils.distill_data import get_synthetic_data  # noqa: E402


def main(argv=None):
    parser = ArgumentParser()
    parser.add_argument("--asr_model", type=str, default="QuartzNet15x5Base-En", required=True)
    parser.add_argument("--dataset", type=str, default=None, help="path to evaluation data (unused by the synthesis itself)")
    parser.add_argument("--num_batch", type=int, default=50, help="number of batches of the synthetic data")
    parser.add_argument("--batch_size", type=int, default=8, help="batch size of the synthetic data")
    parser.add_argument("--seqlen", type=int, default=500, help="sequence length of the synthetic data")
    parser.add_argument("--train_iter", type=int, default=200, help="training iterations for the synthetic data generation")
    parser.add_argument("--dump_path", type=str, default=None, help="path to dump the synthetic data")
    parser.add_argument("--dump_prefix", type=str, default='syn', help="prefix for the filename of the dumped synthetic data")
    parser.add_argument("--lr", type=float, default=0.01, help="Learning rate for the synthetic data generation")
    parser.add_argument("--synthetic_model", action='store_true', help="(extension) random-init weights of --asr_model")
    parser.add_argument("--seed", type=int, default=None, help="(extension) seed of the initial uniform noise")
    parser.add_argument("--cpu", action='store_true', help="(extension) run on the CPU (the reference requires a GPU)")
    args = parser.parse_args(argv)
    if not args.cpu and not torch.cuda.is_available():
        raise Exception("Current implementation only supports GPU (pass --cpu to run the float model on the host)")
    torch.set_grad_enabled(False)
    if args.asr_model.endswith('.nemo'):
        teacher_model = EncDecCTCModel.restore_from(restore_path=args.asr_model)
    elif args.synthetic_model:
        teacher_model = EncDecCTCModel.from_synthetic(args.asr_model)
    else:
        teacher_model = EncDecCTCModel.from_pretrained(model_name=args.asr_model)
    if not args.cpu:
        teacher_model = teacher_model.cuda()
    teacher_model.set_quant_mode('none')                     # the float teacher
    print("Num batches: %d, Batch size: %d, Training iterations: %d, Learning rate: %.3f "
          % (args.num_batch, args.batch_size, args.train_iter, args.lr))
    print('Synthesizing...')
    feat_in = teacher_model.encoder._feat_in
    synthetic_data = get_synthetic_data(teacher_model.encoder, teacher_model.decoder, batch_size=args.batch_size, dim=feat_in,
                                        seqlen=args.seqlen, num_batch=args.num_batch, train_iter=args.train_iter, lr=args.lr,
                                        seed=args.seed)
    file_name = '%s_nb%d_iter%d_lr%.3f.pkl' % (args.dump_prefix, args.num_batch, args.train_iter, args.lr)
    if args.dump_path is not None:
        os.makedirs(args.dump_path, exist_ok=True)
        file_name = os.path.join(args.dump_path, file_name)
    print('Synthetic data dumped as ', file_name)
    with open(file_name, 'wb') as f:
        pickle.dump([x.cpu() for x in synthetic_data], f)
    return file_name


if __name__ == '__main__':
    main()
