"""Transcript cleaning for the English character parser (nemo/collections/asr/parts/cleaners.py:93-215), restated.

The reference delegates to two third-party packages that are not in this image: `unidecode` (ASCII transliteration) and
`inflect` (number -> words).  Both are restated here for the inputs transcripts contain - NFKD accent folding, and
inflect's default English cardinals / ordinals ("one thousand, two hundred and thirty-four", "twenty-first"; the commas
and hyphens then fall to the punctuation table exactly as in the reference).  Parity with inflect's handling of malformed
number strings is unpinned (no fixture of the reference covers it); LibriSpeech transcripts contain neither digits nor
punctuation, so the dev-clean WER path never reaches that code."""
import re
import unicodedata

NUM_CHECK = re.compile(r'([$]?)(^|\s)(\S*[0-9]\S*)(?=(\s|$)((\S*)(\s|$))?)')
TIME_CHECK = re.compile(r'([0-9]{1,2}):([0-9]{2})(am|pm)?')
CURRENCY_CHECK = re.compile(r'\$')
ORD_CHECK = re.compile(r'([0-9]+)(st|nd|rd|th)')
THREE_CHECK = re.compile(r'([0-9]{3})([.,][0-9]{1,2})?([!.?])?$')
DECIMAL_CHECK = re.compile(r'([.,][0-9]{1,2})$')

_ABBR = [("ms", "miss"), ("mrs", "misess"), ("mr", "mister"), ("messrs", "messeurs"), ("dr", "doctor"), ("drs", "doctors"),
         ("st", "saint"), ("co", "company"), ("jr", "junior"), ("sr", "senior"), ("rev", "reverend"), ("hon", "honorable"),
         ("sgt", "sergeant"), ("capt", "captain"), ("maj", "major"), ("col", "colonel"), ("lt", "lieutenant"),
         ("gen", "general"), ("prof", "professor"), ("lb", "pounds"), ("rep", "representative"), ("st", "street"),
         ("ave", "avenue"), ("etc", "et cetera"), ("jan", "january"), ("feb", "february"), ("mar", "march"),
         ("apr", "april"), ("jun", "june"), ("jul", "july"), ("aug", "august"), ("sep", "september"), ("oct", "october"),
         ("nov", "november"), ("dec", "december")]
ABBREVIATIONS_COMMON = [(re.compile('\\b%s\\.' % a), b) for a, b in _ABBR]          # cleaners.py:31-70 (same order)

_ONES = ['zero', 'one', 'two', 'three', 'four', 'five', 'six', 'seven', 'eight', 'nine', 'ten', 'eleven', 'twelve',
         'thirteen', 'fourteen', 'fifteen', 'sixteen', 'seventeen', 'eighteen', 'nineteen']
_TENS = ['', '', 'twenty', 'thirty', 'forty', 'fifty', 'sixty', 'seventy', 'eighty', 'ninety']
_GROUPS = ['', ' thousand', ' million', ' billion', ' trillion', ' quadrillion']
_ORD = {'one': 'first', 'two': 'second', 'three': 'third', 'five': 'fifth', 'eight': 'eighth', 'nine': 'ninth',
        'twelve': 'twelfth'}


def _hundreds(n):
    out = []
    if n >= 100:
        out.append(_ONES[n // 100] + ' hundred')
        n %= 100
        if n:
            out.append('and')
    if n >= 20:
        out.append(_TENS[n // 10] + ('-' + _ONES[n % 10] if n % 10 else ''))
    elif n or not out:
        out.append(_ONES[n])
    return ' '.join(out)


def _cardinal(n):
    """inflect.number_to_words(int) with its defaults (andword='and', comma between groups)."""
    if n == 0:
        return 'zero'
    parts, g = [], 0
    while n:
        n, r = divmod(n, 1000)
        if r:
            parts.append(_hundreds(r) + _GROUPS[min(g, len(_GROUPS) - 1)])
        g += 1
    parts.reverse()
    if len(parts) > 1 and ' and ' not in parts[-1] and ' hundred' not in parts[-1]:
        return ', '.join(parts[:-1]) + ' and ' + parts[-1]   # 1 005 -> "one thousand and five"
    return ', '.join(parts)


def number_to_words(s):
    """The subset of inflect.engine().number_to_words the cleaner calls: digit strings (commas allowed), one decimal
    point ("three point five zero" digit by digit), ordinals ("21st" -> "twenty-first")."""
    s = str(s)
    m = ORD_CHECK.fullmatch(s)
    if m:
        w = _cardinal(int(m.group(1)))
        head, sep, last = w.rpartition('-') if '-' in w.split(' ')[-1] else w.rpartition(' ')
        last_o = _ORD.get(last, last[:-1] + 'ieth' if last.endswith('y') else last + 'th')
        return (head + sep + last_o) if sep else last_o
    digits = re.sub(r'[^0-9.]', '', s)
    if not digits.strip('.'):
        return ''
    whole, _, frac = digits.partition('.')
    words = _cardinal(int(whole)) if whole else 'zero'
    if frac:
        words += ' point ' + ' '.join(_ONES[int(c)] for c in frac if c.isdigit())
    return words


def unidecode(string):
    """ASCII transliteration for accented Latin text (stand-in for the `unidecode` package): NFKD, combining marks dropped."""
    return ''.join(c for c in unicodedata.normalize('NFKD', string) if not unicodedata.combining(c) and ord(c) < 128)


class NumberCleaner:                                          # cleaners.py:132-215
    def __init__(self):
        self.reset()

    def reset(self):
        self.curr_num, self.currency = [], None

    def format_final_number(self, whole_num, decimal):
        if self.currency:
            s = number_to_words(whole_num)
            s += " dollar" if whole_num == 1 else " dollars"     # (the reference compares a str with 1: always plural)
            if decimal:
                s += " and " + number_to_words(decimal)
                s += " cent" if whole_num == decimal else " cents"
            self.reset()
            return s
        self.reset()
        if decimal:
            return number_to_words(whole_num + "." + decimal)
        return re.sub(r'[0-9,]+', lambda m: " " + number_to_words(m.group(0)) + " ", whole_num)

    def clean(self, match):
        ws, number = match.group(2), match.group(3)
        t = TIME_CHECK.match(number)
        if t:
            mins = int(t.group(2))
            return (ws + number_to_words(t.group(1)) + (" " + number_to_words(t.group(2)) if mins else "")
                    + (" " + t.group(3) if t.group(3) else ""))
        o = ORD_CHECK.match(number)
        if o:
            return ws + number_to_words(o.group(0))
        if self.currency is None:
            self.currency = match.group(1) or CURRENCY_CHECK.match(number)
        if THREE_CHECK.match(match.group(6) or ''):
            self.curr_num.append(number)
            return " "
        whole_num = "".join(self.curr_num) + number
        decimal = None
        d = DECIMAL_CHECK.search(whole_num)
        if d:
            decimal = d.group(1)[1:]
            whole_num = whole_num[: -len(decimal) - 1]
        return ws + self.format_final_number(re.sub(r'\.', '', whole_num), decimal)


def clean_numbers(string):
    return NUM_CHECK.sub(NumberCleaner().clean, string)


def clean_abbreviations(string):
    for regex, replacement in ABBREVIATIONS_COMMON:
        string = re.sub(regex, replacement, string)
    return string


def clean_punctuations(string, table, punctuation_to_replace):
    for punc, replacement in punctuation_to_replace.items():
        string = re.sub('\\{}'.format(punc), " {} ".format(replacement), string)
    return string.translate(table)


def clean_text(string, table, punctuation_to_replace):        # cleaners.py:93-102
    string = unidecode(string)
    string = string.lower()
    string = re.sub(r'\s+', " ", string)
    string = clean_numbers(string)
    string = clean_abbreviations(string)
    string = clean_punctuations(string, table, punctuation_to_replace)
    return re.sub(r'\s+', " ", string).strip()
