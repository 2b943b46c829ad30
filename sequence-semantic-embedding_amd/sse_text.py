"""Text -> token-id side of the drop-in boundary: the invertible word splitter
and the subword vocabulary the reference uses (its `tokenizer.py:68-90` and
`text_encoder.py:334-356,427-436,491-519,534-686,717-760`), written from the
published algorithm so that `vocabulary.txt` files and token ids are
interchangeable with the reference's (checked against tests/golden/prep_*.json,
which the reference's own code produced).

Not on the accelerated path (plain CPU string work above the C ABI); it exists so
that the sse_train / sse_index / sse_demo command lines run without the
reference tree.
"""
import collections
import sys
import unicodedata

PAD, EOS = "<pad>", "<EOS>"
RESERVED = [PAD, EOS]
PAD_ID, EOS_ID = 0, 1                     # text_encoder.py:40-45

_ESCAPE_ALPHABET = set("\\_u;0123456789")


def _is_word_char(ch, _cache={}):
    v = _cache.get(ch)
    if v is None:
        v = unicodedata.category(ch)[0] in "LN"     # letters and numbers (tokenizer.py:62-65)
        _cache[ch] = v
    return v


def split_tokens(text):
    """Split at every alphanumeric/non-alphanumeric boundary; a lone space between
    two words is dropped (it is implied), except at the very start."""
    if not text:
        return []
    out, start = [], 0
    prev = _is_word_char(text[0])
    for i in range(1, len(text)):
        cur = _is_word_char(text[i])
        if cur != prev:
            piece = text[start:i]
            if piece != " " or start == 0:
                out.append(piece)
            start, prev = i, cur
    out.append(text[start:])
    return out


def join_tokens(tokens):
    out = []
    for i, tok in enumerate(tokens):
        if i and _is_word_char(tokens[i - 1][0]) and _is_word_char(tok[0]):
            out.append(" ")
        out.append(tok)
    return "".join(out)


def escape_token(token, alphabet):
    """Make a token expressible as a concatenation of subtokens: backslash and
    underscore are escaped, characters outside the alphabet become \\<ord>;, and a
    terminating underscore is appended."""
    token = token.replace("\\", "\\\\").replace("_", "\\u")
    return "".join(c if (c in alphabet and c != "\n") else "\\%d;" % ord(c) for c in token) + "_"


class SubwordVocab(object):
    """Greedy longest-match subword vocabulary (the reference's SubwordTextEncoder)."""

    def __init__(self, filename=None):
        self.subtokens = []
        self._ids = {}
        self._maxlen = 0
        self.alphabet = set()
        if filename is not None:
            self.load(filename)

    # -- vocabulary file: one subtoken per line, optionally quoted (text_encoder.py:731-760)
    def load(self, filename):
        toks = []
        with open(filename, encoding="utf-8") as f:
            for line in f:
                s = line.strip()
                if len(s) >= 2 and ((s[0] == "'" and s[-1] == "'") or (s[0] == '"' and s[-1] == '"')):
                    s = s[1:-1]
                toks.append(s)
        self._set_subtokens(toks, reserved=0)
        self.alphabet = {c for t in toks for c in t} | _ESCAPE_ALPHABET

    def store(self, filename):
        with open(filename, "w", encoding="utf-8") as f:
            for s in self.subtokens:
                f.write("'" + s + "'\n")

    def _set_subtokens(self, toks, reserved):
        self.subtokens = (RESERVED + list(toks)) if reserved else list(toks)
        self._maxlen = max(len(s) for s in toks)
        self._ids = {s: i + reserved for i, s in enumerate(toks) if s}

    @property
    def vocab_size(self):
        return len(self.subtokens)

    # -- encoding
    def _segment(self, escaped):
        pieces, pos, n = [], 0, len(escaped)
        while pos < n:
            for end in range(min(n, pos + self._maxlen), pos, -1):
                if escaped[pos:end] in self._ids:
                    pieces.append(escaped[pos:end])
                    pos = end
                    break
            else:
                raise AssertionError("token not encodable with this vocabulary: %r" % escaped)
        return pieces

    def encode(self, text):
        ids = []
        for tok in split_tokens(text):
            ids.extend(self._ids[p] for p in self._segment(escape_token(tok, self.alphabet)))
        return ids

    def decode(self, ids):
        import re
        text = "".join(self.subtokens[i] if 0 <= i < len(self.subtokens) else "" for i in ids)

        def unesc(m):
            if m.group(1) is None:
                return "_" if m.group(0) == "\\u" else "\\"
            try:
                return chr(int(m.group(1)))
            except (ValueError, OverflowError):
                return ""

        toks = [re.sub(r"\\u|\\\\|\\([0-9]+);", unesc, t) for t in text.split("_") if t]
        return join_tokens(toks)

    # -- building (text_encoder.py:534-686)
    def _build_once(self, token_counts, min_count, iterations):
        self.alphabet = {c for t in token_counts for c in t} | {c for t in RESERVED for c in t} | _ESCAPE_ALPHABET
        self._set_subtokens(sorted(self.alphabet), reserved=len(RESERVED))
        min_count = max(1, min_count)
        for _ in range(iterations):
            counts = collections.defaultdict(int)
            for token, cnt in token_counts.items():
                esc = escape_token(token, self.alphabet)
                pos = 0
                for piece in self._segment(esc):
                    for end in range(pos + 1, len(esc) + 1):
                        counts[esc[pos:end]] += cnt
                    pos += len(piece)
            by_len = collections.defaultdict(list)
            for s, c in counts.items():
                if c >= min_count:
                    by_len[len(s)].append(s)
            chosen = []
            for length in sorted(by_len, reverse=True):
                for s in by_len[length]:
                    c = counts[s]
                    if c >= min_count:
                        if s not in self.alphabet:
                            chosen.append((c, s))
                        for l in range(1, length):
                            counts[s[:l]] -= c
            chosen.extend((counts.get(a, 0), a) for a in self.alphabet)
            chosen.sort(reverse=True)
            self._set_subtokens([s for _, s in chosen], reserved=len(RESERVED))

    @classmethod
    def build_to_target_size(cls, target_size, token_counts, min_val, max_val, iterations=4, log=None):
        """Bisection on the minimum subtoken count until the vocabulary size is within
        1 % of `target_size` (or the interval is exhausted); the closer of the
        candidates wins."""
        if min_val > max_val:
            raise ValueError("Lower bound for the minimum token count is greater than the upper bound.")
        if target_size < 1:
            raise ValueError("Target size must be positive.")

        def search(lo, hi):
            mid = (lo + hi) // 2
            cand = cls()
            cand._build_once(token_counts, mid, iterations)
            if log:
                log("vocab size %d at min_count %d" % (cand.vocab_size, mid))
            close_enough = abs(cand.vocab_size - target_size) * 100 < target_size
            if close_enough or lo >= hi or mid < 2:
                return cand
            other = search(mid + 1, hi) if cand.vocab_size > target_size else search(lo, mid - 1)
            if other is not None and abs(other.vocab_size - target_size) < abs(cand.vocab_size - target_size):
                return other
            return cand

        return search(min_val, max_val)


def corpus_token_counts(paths, max_lines):
    """Token counts over the lines of the given files (tokenizer.py:150-171)."""
    counts, seen = collections.Counter(), 0
    for path in sorted(paths):
        with open(path, encoding="utf-8") as f:
            for line in f:
                counts.update(split_tokens(line.strip()))
                seen += 1
                if max_lines and seen >= max_lines:
                    return counts
    return counts


def pad_tokens(tokens, max_seq_length):
    """Left-pad with PAD, close with EOS; over-long inputs keep T-2 tokens
    (sse_index.py:79-85, data_utils.py:149-155,194-198)."""
    tokens = list(tokens)
    if len(tokens) > max_seq_length - 2:
        return [PAD_ID] + tokens[:max_seq_length - 2] + [EOS_ID]
    return [PAD_ID] * (max_seq_length - len(tokens) - 1) + tokens + [EOS_ID]


if __name__ == "__main__":
    v = SubwordVocab(sys.argv[1])
    for line in sys.stdin:
        print(v.encode(line.strip().lower()))
