"""The term lookup of the keyword chain (orr_kernels.hip: vocab_match_lookup_kernel), restated on the CPU.

KeywordScore (RecallSearchService.cs:111) asks whether a query term occurs in a chunk's lower-cased content; the library
answers through the vocabulary: which TOKENS contain which term, then the tokens' posting lists.  For many terms the
device does not compare every (token, term) pair: the host sorts the terms by their length-masked first four bytes in
four classes (1, 2, 3 and 4+ bytes), a lane looks each window of its token up in every class (Bloom filter, binary search,
run of equal keys), verifies the candidates and reports a (token, term) pair once -- at the first start where the term
occurs.  This file restates exactly that with Python integers and checks it against `term in token` on random and
engineered vocabularies (repeated letters, terms that occur several times in a token, terms that are prefixes or
suffixes of one another, one-letter terms, 16-byte tokens and terms).
(The device code itself is compared with the oracle in tests/test_gpu_parity.py, with the lookup forced in
profiles/r03_switch_matrix.txt.)
"""
import itertools
import random

BLOOM_BITS = 1 << 17


def bloom_hash(key, cls):
    h = ((key ^ ((cls * 0x9E3779B9) & 0xFFFFFFFF)) * 2654435761) & 0xFFFFFFFF
    h ^= h >> 15
    return h & (BLOOM_BITS - 1)


def first_dword(b):
    """The first four bytes little-endian, missing bytes zero (MatchTerm.w[0] under its mask)."""
    return int.from_bytes(b[:4].ljust(4, b"\0"), "little")


def build_tables(terms):
    """[(class, key, term number)] sorted, class boundaries, Bloom bits -- what launch_keyword_side uploads."""
    order = sorted((min(len(t), 4) - 1, first_dword(t), i) for i, t in enumerate(terms) if 1 <= len(t) <= 16)
    lk = [0] * 5
    for pos, (c, _, _) in enumerate(order):
        lk[c + 1] = pos + 1
    for c in range(1, 5):
        lk[c] = max(lk[c], lk[c - 1])
    bloom = set(bloom_hash(k, c) for c, k, _ in order)
    return order, lk, bloom


def term_at(padded, i, term):
    return padded[i:i + len(term)] == term


def lookup_hits(token, terms, order, lk, bloom):
    """The kernel's walk for one token of at most 16 bytes: the set of term numbers it reports, and how often each."""
    padded = token + b" " * 32                                    # the pool pads with spaces, which no term contains
    reported = []
    for c in range(4):
        c0, c1 = lk[c], lk[c + 1]
        if c0 == c1:
            continue
        mask = (1 << (8 * (c + 1))) - 1 if c < 3 else 0xFFFFFFFF
        for i in range(16):
            if i + c + 1 > len(token):
                continue
            key = int.from_bytes(padded[i:i + 4], "little") & mask
            if bloom_hash(key, c) not in bloom:
                continue
            lo, hi = c0, c1
            while lo < hi:
                mid = (lo + hi) // 2
                if order[mid][1] < key:
                    lo = mid + 1
                else:
                    hi = mid
            pos = lo
            while pos < c1 and order[pos][1] == key:
                t = order[pos][2]
                pos += 1
                term = terms[t]
                if len(term) > len(token) - i or not term_at(padded, i, term):
                    continue
                if any(term_at(padded, e, term) for e in range(i)):
                    continue                                      # an earlier start reports the pair
                reported.append(t)
    return reported


def check(tokens, terms):
    order, lk, bloom = build_tables(terms)
    for tok in tokens:
        got = lookup_hits(tok, terms, order, lk, bloom)
        want = sorted(i for i, t in enumerate(terms) if 1 <= len(t) <= 16 and t in tok)
        assert sorted(got) == want, (tok, [terms[i] for i in got], [terms[i] for i in want])
        assert len(got) == len(set(got)), (tok, got)              # one hit per (token, term)


def test_random_vocabulary_and_substring_terms():
    rng = random.Random(5)
    syll = [b"ka", b"re", b"mi", b"to", b"ne", b"su", b"lo", b"vi", b"da", b"po", b"er", b"in", b"a", b"e", b"x1", b"\xc3\xa9"]
    tokens = list({b"".join(rng.choice(syll) for _ in range(rng.randint(1, 8)))[:16] for _ in range(3000)})
    terms = set()
    for _ in range(700):                                          # substrings of tokens, whole tokens, and strangers
        t = rng.choice(tokens)
        a = rng.randrange(len(t))
        terms.add(t[a:a + rng.randint(1, 16)])
    terms |= {b"zzz", b"q", b"kareka", b"ererer", rng.choice(tokens)}
    terms = [t for t in terms if t]
    check(tokens, terms)


def test_engineered_cases():
    tokens = [b"aaaaaaaaaaaaaaaa", b"abababababababab", b"a", b"ab", b"abc", b"abcd", b"abcde", b"abcdefghijklmnop",
              b"xabcdefghijklmno", b"kubernetes", b"netnetnet", b"helm", b"mlehhelm", b"zzzzzzzzzzzzzzzz", b"0123456789abcdef"]
    terms = [b"a", b"aa", b"aaa", b"aaaa", b"aaaaa", b"aaaaaaaaaaaaaaaa", b"ab", b"aba", b"abab", b"b", b"ba", b"abc", b"abcd", b"abcde",
             b"bcde", b"cde", b"de", b"e", b"abcdefghijklmnop", b"bcdefghijklmnop", b"mnop", b"nop", b"op", b"p", b"net", b"tne", b"etn",
             b"helm", b"elm", b"lm", b"m", b"hhelm", b"zzzz", b"zzzzzzzzzzzzzzzz", b"zzzzzzzzzzzzzzzzz", b"0123", b"cdef", b"9abc", b"",
             b"abcdefghijklmnopq", b"kubernetes", b"kubernetesx", b"ubernete"]
    terms = [t for t in terms]
    check(tokens, terms)


def test_every_split_of_short_alphabets():
    """Exhaustive over a two-letter alphabet: every token of up to 7 bytes against every term of up to 5."""
    alpha = [b"a", b"b"]
    tokens = [b"".join(p) for n in range(1, 8) for p in itertools.product(alpha, repeat=n)]
    terms = [b"".join(p) for n in range(1, 6) for p in itertools.product(alpha, repeat=n)]
    check(tokens, terms)
