"""Character parsers (nemo/collections/asr/parts/parsers.py:24-176): text -> label ids for the evaluation transcripts."""
import string
from typing import List, Optional

from . import cleaners


class CharParser:
    """parsers.py:24-98: optional strip / lower-case normalisation, per-character lookup, multi-character labels as
    whole words, unknown characters mapped to `unk_id` and dropped when `unk_id == blank_id`."""

    def __init__(self, labels: List[str], *, unk_id: int = -1, blank_id: int = -1, do_normalize: bool = True,
                 do_lowercase: bool = True):
        self._labels = labels
        self._unk_id, self._blank_id = unk_id, blank_id
        self._do_normalize, self._do_lowercase = do_normalize, do_lowercase
        self._labels_map = {label: index for index, label in enumerate(labels)}
        self._special_labels = set(label for label in labels if len(label) > 1)

    def __call__(self, text: str) -> Optional[List[int]]:
        if self._do_normalize:
            text = self._normalize(text)
            if text is None:
                return None
        return self._tokenize(text)

    def _normalize(self, text: str) -> Optional[str]:
        text = text.strip()
        return text.lower() if self._do_lowercase else text

    def _tokenize(self, text: str) -> List[int]:
        tokens = []
        for word_id, word in enumerate(text.split(' ')):
            if word_id != 0:
                tokens.append(self._labels_map.get(' ', self._unk_id))
            if word in self._special_labels:
                tokens.append(self._labels_map[word])
                continue
            tokens.extend(self._labels_map.get(char, self._unk_id) for char in word)
        return [t for t in tokens if t != self._blank_id]


class ENCharParser(CharParser):
    """parsers.py:101-145: English cleaning (cleaners.clean_text) with a punctuation table that maps every punctuation
    character that is neither replaced by a word (+ & %) nor a label to a space."""
    PUNCTUATION_TO_REPLACE = {'+': 'plus', '&': 'and', '%': 'percent'}

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        punctuation = string.punctuation
        for char in self.PUNCTUATION_TO_REPLACE:
            punctuation = punctuation.replace(char, '')
        for label in self._labels:
            punctuation = punctuation.replace(label, '')
        self._table = str.maketrans(punctuation, ' ' * len(punctuation))

    def _normalize(self, text: str) -> Optional[str]:
        try:
            return cleaners.clean_text(string=text, table=self._table, punctuation_to_replace=self.PUNCTUATION_TO_REPLACE)
        except Exception:
            return None


NAME_TO_PARSER = {'base': CharParser, 'en': ENCharParser}


def make_parser(labels: Optional[List[str]] = None, name: str = 'base', **kwargs) -> CharParser:
    """parsers.py:151-176."""
    if name not in NAME_TO_PARSER:
        raise ValueError('Invalid parser name.')
    if labels is None:
        labels = list(string.printable)
    return NAME_TO_PARSER[name](labels=labels, **kwargs)
