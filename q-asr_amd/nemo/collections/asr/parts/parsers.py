"""Transcript -> label ids for the evaluation transcripts (behaviour of nemo/collections/asr/parts/parsers.py:24-176).

Contract kept for the drop-in path (`make_parser(labels, 'en', unk_id=-1, blank_id=-1, do_normalize=...)` as
AudioToCharDataset calls it, audio_to_text.py:258-267):
  * `CharParser`: optional normalisation (strip, lower-case), then the text is cut at single spaces; a word that IS a
    multi-character label becomes that label's id, any other word one id per character; words are joined by the id of
    ' '; characters (or a ' ') that are not labels become `unk_id`; ids equal to `blank_id` are dropped at the end.
  * `ENCharParser`: normalisation = `cleaners.clean_text` with a translation table that turns every ASCII punctuation
    mark into a space, except the marks spoken as words ('+' plus, '&' and, '%' percent) and anything that is a label;
    a transcript the cleaner cannot handle yields None (the dataset skips it).
  * `make_parser(labels=None, name='base', **kwargs)`: 'base' | 'en'; default labels = string.printable.
Parity: the reference module cannot be imported here (frozendict / inflect are absent) - the cases of
tests/test_facade_cpu.py::test_en_char_parser are worked by hand from the reference's documented behaviour
("parity unpinned")."""
import string
from typing import Dict, List, Optional, Sequence

from . import cleaners

SPOKEN_PUNCTUATION = {'+': 'plus', '&': 'and', '%': 'percent'}
WORD_SEPARATOR = ' '


class CharParser:
    def __init__(self, labels: Sequence[str], *, unk_id: int = -1, blank_id: int = -1, do_normalize: bool = True,
                 do_lowercase: bool = True):
        self.labels = list(labels)
        self.unk_id, self.blank_id = unk_id, blank_id
        self.do_normalize, self.do_lowercase = do_normalize, do_lowercase
        self._id_of: Dict[str, int] = {}
        for position, label in enumerate(self.labels):
            self._id_of[label] = position                      # a repeated label keeps its LAST position, like a dict build
        self._word_labels = frozenset(label for label in self.labels if len(label) > 1)

    # -- normalisation -------------------------------------------------------------------------------------------
    def normalize(self, text: str) -> Optional[str]:
        stripped = text.strip()
        return stripped.lower() if self.do_lowercase else stripped

    def _normalize(self, text: str) -> Optional[str]:          # (the reference's private name for it)
        return self.normalize(text)

    # -- ids -----------------------------------------------------------------------------------------------------
    def _word_ids(self, word: str) -> List[int]:
        if word in self._word_labels:
            return [self._id_of[word]]
        lookup, unknown = self._id_of.get, self.unk_id
        return [lookup(ch, unknown) for ch in word]

    def encode(self, text: str) -> List[int]:
        separator = self._id_of.get(WORD_SEPARATOR, self.unk_id)
        ids: List[int] = []
        for position, word in enumerate(text.split(WORD_SEPARATOR)):   # consecutive spaces give empty words: one separator each
            if position:
                ids.append(separator)
            ids += self._word_ids(word)
        blank = self.blank_id
        return [i for i in ids if i != blank]

    def __call__(self, text: str) -> Optional[List[int]]:
        if self.do_normalize:
            text = self.normalize(text)
            if text is None:
                return None
        return self.encode(text)


class ENCharParser(CharParser):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        to_space = ''.join(ch for ch in string.punctuation if ch not in SPOKEN_PUNCTUATION and ch not in self._id_of)
        for label in self._word_labels:                        # a multi-character label made of punctuation keeps its marks
            to_space = to_space.replace(label, '')
        self._punctuation_table = str.maketrans(to_space, ' ' * len(to_space))

    def normalize(self, text: str) -> Optional[str]:
        try:
            return cleaners.clean_text(string=text, table=self._punctuation_table, punctuation_to_replace=SPOKEN_PUNCTUATION)
        except Exception:                                      # the dataset drops transcripts the cleaner rejects
            return None


PARSERS = {'base': CharParser, 'en': ENCharParser}
NAME_TO_PARSER = PARSERS                                      # the reference's name for the registry


def make_parser(labels: Optional[Sequence[str]] = None, name: str = 'base', **kwargs) -> CharParser:
    try:
        cls = PARSERS[name]
    except KeyError:
        raise ValueError('Invalid parser name.') from None
    return cls(labels=list(string.printable) if labels is None else labels, **kwargs)
