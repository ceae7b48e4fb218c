"""Greedy CTC decoding and word error rate (nemo/collections/asr/metrics/wer.py:26-59, 62-181).
The reference delegates the Levenshtein distance to the third-party `editdistance==0.5.3` C extension
(absent here); a dynamic-programming restatement is used and pinned by the reference's own known
answers (tests/collections/asr/test_asr_metrics.py:94-111)."""
from typing import List

import torch

__all__ = ['word_error_rate', 'WER']


def _levenshtein(a, b) -> int:
    if len(a) < len(b):
        a, b = b, a
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i] + [0] * len(b)
        for j, y in enumerate(b, 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y))
        prev = cur
    return prev[-1]


def word_error_rate(hypotheses: List[str], references: List[str], use_cer=False) -> float:
    if len(hypotheses) != len(references):
        raise ValueError("In word error rate calculation, hypotheses and reference lists must have the same "
                         "number of elements. But I got:{0} and {1} correspondingly".format(len(hypotheses),
                                                                                          len(references)))
    scores = words = 0
    for h, r in zip(hypotheses, references):
        h_list, r_list = (list(h), list(r)) if use_cer else (h.split(), r.split())
        words += len(r_list)
        scores += _levenshtein(h_list, r_list)
    return 1.0 * scores / words if words != 0 else float('inf')


class WER:
    """Accumulating WER metric with the reference's decode helpers (no pytorch-lightning dependency)."""

    def __init__(self, vocabulary, batch_dim_index=0, use_cer=False, ctc_decode=True, log_prediction=True,
                 dist_sync_on_step=False):
        self.batch_dim_index = batch_dim_index
        self.blank_id = len(vocabulary)
        self.labels_map = dict(enumerate(vocabulary))
        self.use_cer, self.ctc_decode, self.log_prediction = use_cer, ctc_decode, log_prediction
        self.scores = torch.tensor(0)
        self.words = torch.tensor(0)

    def ctc_decoder_predictions_tensor(self, predictions: torch.Tensor) -> List[str]:
        """Collapse repeats, drop blanks; walks the full padded row like the reference (it ignores lengths)."""
        rows = predictions.long().cpu()
        if self.batch_dim_index != 0:
            rows = rows.transpose(0, self.batch_dim_index)
        hyps = []
        for row in rows.tolist():
            out, prev = [], self.blank_id
            for p in row:
                if (p != prev or prev == self.blank_id) and p != self.blank_id:
                    out.append(p)
                prev = p
            hyps.append(''.join(self.labels_map[c] for c in out))
        return hyps

    def update(self, predictions, targets, target_lengths):
        refs = []
        t = targets.long().cpu()
        n = target_lengths.long().cpu()
        for i in range(t.shape[0]):
            refs.append(''.join(self.labels_map[c] for c in t[i][:int(n[i])].tolist()))
        if not self.ctc_decode:
            raise NotImplementedError("Not supported. Use BeamSearch in the meantime")
        hyps = self.ctc_decoder_predictions_tensor(predictions)
        scores = words = 0
        for h, r in zip(hyps, refs):
            h_l, r_l = (list(h), list(r)) if self.use_cer else (h.split(), r.split())
            words += len(r_l)
            scores += _levenshtein(h_l, r_l)
        self.scores = self.scores + scores
        self.words = self.words + words

    def compute(self):
        s, w = self.scores.detach().float(), self.words.detach().float()
        return s / w, s, w

    def reset(self):
        self.scores, self.words = torch.tensor(0), torch.tensor(0)
