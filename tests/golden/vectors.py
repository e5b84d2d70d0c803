"""Golden vectors transcribed BY HAND from the reference's own unit tests.

The reference (simdjson-java) is pure Java and cannot be executed in this image (no JVM),
so these literals -- copied from the assertions in /root/reference/src/test/java/org/simdjson/
-- are what pins the CPU oracle (and, through it, the HIP kernels).  Every entry cites the
test file:line it comes from.  Inputs are bytes; `idx` is the exact list of structural
indexes the reference asserts (via BitIndexes.getAndAdvance until isEnd()).
"""

A49 = "a" * 49
A51 = "a" * 51

# ---------------------------------------------------------------------------------------------
# StructuralIndexerTest.java
# ---------------------------------------------------------------------------------------------
STRUCTURAL_INDEXER = [
    # (name, input bytes, expected indexes or None, expected error message or None, cite)
    ("unquotedString", b"abc 123", [0, 4], None, "StructuralIndexerTest.java:14-29"),
    ("quotedString", b'"abc 123"', [0], None, "StructuralIndexerTest.java:31-45"),
    ("unclosedString", b'"abc 123', None, "Unclosed string. A string is opened, but never closed.",
     "StructuralIndexerTest.java:47-63"),
    ("quotedStringSpanningMultipleBlocks",
     b'abc "a0 a1 a2 a3 a4 a5 a6 a7 a8 a9 b0 b1 b2 b3 b4 b5 b6 b7 b8 b9 c0 c1 c2 c3 c4 c5 c6 c7 c8 c9 '
     b'd0 d1 d2 d3 d4 d5 d6 d7 d8 d" def',
     [0, 4, 125], None, "StructuralIndexerTest.java:65-81"),
    ("escapedQuote[1]", b'abc \\"123', [0, 4], None, "StructuralIndexerTest.java:83-101"),
    ("escapedQuote[2]", b'abc \\\\\\"123', [0, 4], None, "StructuralIndexerTest.java:83-101"),
    ("escapedQuoteSpanningMultipleBlocks",
     b'a0ba1ca2ca3ca4ca5ca6ca7ca8ca9cb0cb1cb2cb3cb4cb5cb6cb7cb8cb9cc0 \\"def',
     [0, 63], None, "StructuralIndexerTest.java:103-118"),
    ("unescapedQuote[1]", b'abc \\\\"123', None, "Unclosed string. A string is opened, but never closed.",
     "StructuralIndexerTest.java:120-139"),
    ("unescapedQuote[2]", b'abc \\\\\\\\"123', None, "Unclosed string. A string is opened, but never closed.",
     "StructuralIndexerTest.java:120-139"),
    ("unescapedQuoteSpanningMultipleBlocks",
     b'a0 a1 a2 a3 a4 a5 a6 a7 a8 a9 b0 b1 b2 b3 b4 b5 b6 b7 b8 b9 c0 \\\\"abc',
     None, "Unclosed string. A string is opened, but never closed.", "StructuralIndexerTest.java:141-157"),
    ("operatorsClassification", ("a{bc}1:2,3[efg]" + A49).encode(),
     [0, 1, 2, 4, 5, 6, 7, 8, 9, 10, 11, 14, 15], None, "StructuralIndexerTest.java:159-185"),
    ("controlCharactersClassification", b"aaa\x1aa\x0caa" + b"a" * 56,
     [0, 3, 4, 5, 6], None, "StructuralIndexerTest.java:187-214"),
    ("whitespacesClassification", ("a bc\t1\n2\r3efg" + A51).encode(),
     [0, 2, 5, 7, 9], None, "StructuralIndexerTest.java:216-234"),
    ("emptyInput", b"", [], None, "StructuralIndexerTest.java:262-273"),
] + [
    ("inputLengthCloseToVectorWidth[%d]" % n, b"a" * n, [0], None, "StructuralIndexerTest.java:236-260")
    for n in (15, 16, 17, 31, 32, 33, 63, 64, 65)
]
assert len(("a{bc}1:2,3[efg]" + A49)) == 64 and len("a bc\t1\n2\r3efg" + A51) == 64

# ---------------------------------------------------------------------------------------------
# Utf8ValidationTest.java -- invalid sequences embedded in (or appended to) valid UTF-8.
# The reference picks the surrounding text at random (unseeded); tests here seed it.
# ---------------------------------------------------------------------------------------------
UTF8_ERROR = "The input is not valid UTF-8"

UTF8_INVALID_MID = [  # randomUtf8ByteArrayIncluding(...)
    ("twoByteSequenceWithTwoContinuationBytes", bytes([0b11000010, 0b10000000, 0b10000000]), "Utf8ValidationTest.java:71-88"),
    ("twoByteSequenceWithoutContinuationBytes", bytes([0b11000010]), "Utf8ValidationTest.java:90-103"),
    ("threeByteSequenceWithThreeContinuationBytes", bytes([0b11100000, 0b10100000, 0b10000000, 0b10000000]), "Utf8ValidationTest.java:122-140"),
    ("threeByteSequenceWithOneContinuationByte", bytes([0b11100000, 0b10100000]), "Utf8ValidationTest.java:142-157"),
    ("threeByteSequenceWithoutContinuationBytes", bytes([0b11100000]), "Utf8ValidationTest.java:159-172"),
    ("fourByteSequenceWithFourContinuationBytes", bytes([0b11110000, 0b10010000, 0b10000000, 0b10000000, 0b10000000]), "Utf8ValidationTest.java:207-226"),
    ("fourByteSequenceWithTwoContinuationBytes", bytes([0b11110000, 0b10010000, 0b10000000]), "Utf8ValidationTest.java:228-245"),
    ("fourByteSequenceWithOneContinuationByte", bytes([0b11110000, 0b10010000]), "Utf8ValidationTest.java:247-263"),
    ("fourByteSequenceWithoutContinuationBytes", bytes([0b11110000]), "Utf8ValidationTest.java:265-278"),
]
UTF8_INVALID_END = [  # randomUtf8ByteArrayEndedWith(...)
    ("twoByteSequenceWithoutContinuationBytesAtTheEnd", bytes([0b11000010]), "Utf8ValidationTest.java:105-120"),
    ("threeByteSequenceWithOneContinuationByteAtTheEnd", bytes([0b11100000, 0b10100000]), "Utf8ValidationTest.java:174-190"),
    ("threeByteSequenceWithoutContinuationBytesAtTheEnd", bytes([0b11100000]), "Utf8ValidationTest.java:192-205"),
    ("fourByteSequenceWithTwoContinuationBytesAtTheEnd", bytes([0b11110000, 0b10010000, 0b10000000]), "Utf8ValidationTest.java:280-297"),
    ("fourByteSequenceWithOneContinuationByteAtTheEnd", bytes([0b11110000, 0b10010000]), "Utf8ValidationTest.java:299-315"),
    ("fourByteSequenceWithoutContinuationBytesAtTheEnd", bytes([0b11110000]), "Utf8ValidationTest.java:317-330"),
]


def utf8_sequences(frm, to, length):
    """testutils/Utf8TestData.java:18-33 -- (over)long encodings of code points [frm,to]."""
    out = []
    lead = {2: 0xC0, 3: 0xE0, 4: 0xF0}[length]
    for cp in range(frm, to + 1):
        b = bytearray(length)
        cur = cp
        for k in range(length - 1, 0, -1):
            b[k] = 0x80 | (cur & 0x3F)
            cur >>= 6
        b[0] = (lead | (cur & 0x1F)) & 0xFF
        out.append(bytes(b))
    return out


# families asserted invalid one by one in the reference:
UTF8_INVALID_FAMILIES = [
    ("invalidAscii", [bytes([b]) for b in range(128, 256)], "Utf8ValidationTest.java:37-52"),
    ("continuationByteWithoutPrecedingLeadingByte", [bytes([b]) for b in range(0x80, 0xC0)], "Utf8ValidationTest.java:54-69"),
    ("overlongTwoByteSequence", utf8_sequences(0x0000, 0x007F, 2), "Utf8ValidationTest.java:331-348"),
    ("overlongThreeByteSequence", utf8_sequences(0x0000, 0x07FF, 3), "Utf8ValidationTest.java:350-367"),
    ("surrogateCodePoints", utf8_sequences(0xD800, 0xDFFF, 3), "Utf8ValidationTest.java:369-386"),
    ("overlongFourByteSequence", utf8_sequences(0x0000, 0xFFFF, 4), "Utf8ValidationTest.java:388-405"),
    ("tooLargeFourByteSequence", utf8_sequences(0x110000, 0x110400, 4), "Utf8ValidationTest.java:407-424"),
]

# ---------------------------------------------------------------------------------------------
# StringParsingTest.java -- (json text, expected python str or None, expected message or None)
# ---------------------------------------------------------------------------------------------
MSG_TRAILING = "More than one JSON value at the root of the document, or extra characters at the end of the JSON!"
MSG_LOW_RESERVED = "Invalid code point. The range U+DC00–U+DFFF is reserved for low surrogate."
MSG_LOW_NO_U = "Low surrogate should start with '\\u'"
MSG_LOW_RANGE = "Invalid code point. Low surrogate should be in the range U+DC00–U+DFFF."
MSG_BAD_UNICODE = "Invalid unicode escape sequence."
MSG_ESCAPE = "Escaped unexpected character: "
MSG_UNESCAPED = "Unescaped characters. Within strings, there are characters that should be escaped."
MSG_UNCLOSED = "Unclosed string. A string is opened, but never closed."

STRING_ERRORS = [
    ('"abc",', MSG_TRAILING, "StringParsingTest.java:36-49"),
    ('"abc"def', MSG_TRAILING, "StringParsingTest.java:36-49"),
] + [
    ('"%s"' % s, MSG_LOW_NO_U, "StringParsingTest.java:94-107")
    for s in ["\\uD8001", "\\uD800\\1", "\\uD800u", "\\uD800\\e", "\\uD800\\DC00", "\\uD800"]
] + [
    ('"\\uD800\\u"', MSG_LOW_RANGE, "StringParsingTest.java:109-122"),
] + [
    ('"%s"' % s, MSG_BAD_UNICODE, "StringParsingTest.java:146-159") for s in ["\\u", "\\u1", "\\u12", "\\u123"]
] + [
    ('["\\g"]', MSG_ESCAPE, "StringParsingTest.java:161-174"),
    ('["\\ą"]', MSG_ESCAPE, "StringParsingTest.java:161-174"),
    ('"\""', MSG_UNCLOSED, "StringParsingTest.java:229-242"),
    ('"\\"', MSG_UNCLOSED, "StringParsingTest.java:229-242"),
]

LONG_STRING = ('["' + "a" * 70 + '"]', "a" * 70, "StringParsingTest.java:176-191")
ARRAY_OF_STRINGS = ('["abc", "ab\\\\c"]', ["abc", "ab\\c"], "StringParsingTest.java:244-260")
LEN_SHORTER = ('"aaaaa"', 6, MSG_UNCLOSED, "StringParsingTest.java:262-274")

# ---------------------------------------------------------------------------------------------
# survey cross-checks (derived from a model that reproduces every
# StructuralIndexerTest vector) + BenchmarkCorrectnessTest.java:19-42
# ---------------------------------------------------------------------------------------------
FILES = {
    # name: (bytes, structurals, stage-1 status, utf8 valid, first indexes, last indexes)
    "twitter.json": (631515, 55263, 0, True, [0, 4, 14, 16, 22, 30, 40, 42, 52, 65, 67, 75],
                     [631503, 631505, 631511, 631513]),
    "github_events.json": (65132, 4656, 0, True, None, None),
    "wide_bench.json": (166504, 7885, 0, True, None, None),
}
TWITTER_DEFAULT_PROFILE_USERS = 86  # BenchmarkCorrectnessTest.java:40
MALFORMED_FIRST_BAD_OFFSET = 4461   # Utf8ValidationTest.java:436-448 (file must fail); offset from SURVEY.md 2b

# ---------------------------------------------------------------------------------------------
# Array/Object/Boolean/Null ParsingTest.java -- grammar errors with their exact messages.
# (json text, passed length or None for full, expected message or None if it must parse, cite)
# ---------------------------------------------------------------------------------------------
MSG_NO_COMMA_ARRAY = "Missing comma between array values"
MSG_UNRECOGNIZED = "Unrecognized primitive. Expected: string, number, 'true', 'false' or 'null'."
MSG_UNCLOSED_ARRAY = "Unclosed array. Missing ']' for starting '['."

GRAMMAR = [
    ("[1 1]", None, MSG_NO_COMMA_ARRAY, "ArrayParsingTest.java:97-109"),
] + [
    (s, None, MSG_UNRECOGNIZED, "ArrayParsingTest.java:111-124") for s in ["[1,,1]", "[,]", "[,,]"]
] + [
    (s, None, MSG_UNCLOSED_ARRAY, "ArrayParsingTest.java:126-139") for s in ["[,", "[1 ", "[,,", "[1,", "[1", "["]
] + [
    ("[[]]", 3, MSG_NO_COMMA_ARRAY, "ArrayParsingTest.java:141-157"),
    ("[]", 1, MSG_UNCLOSED_ARRAY, "ArrayParsingTest.java:141-157"),
    ('{"field": [1 1]}', None, MSG_NO_COMMA_ARRAY, "ArrayParsingTest.java:159-171"),
] + [
    ('{"field": %s}' % s, None, m, "ArrayParsingTest.java:173-193") for s, m in [
        ("[,", MSG_UNRECOGNIZED), ("[1 ", MSG_NO_COMMA_ARRAY), ("[,,", MSG_UNRECOGNIZED),
        ("[1,", MSG_UNRECOGNIZED), ("[1", MSG_NO_COMMA_ARRAY), ("[", MSG_UNRECOGNIZED)]
] + [
    ("[[[[", 2, MSG_UNCLOSED_ARRAY, "ArrayParsingTest.java:214-226"),
    ("[][[[[", 2, None, "ArrayParsingTest.java:228-245"),
    ("{\\null: 1}", None, "Object does not start with a key", "ObjectParsingTest.java:99-111"),
    ("", None, "No structural element found.", "ObjectParsingTest.java:136-148"),
    ('{"a":{}}', 7, "No comma between object fields", "ObjectParsingTest.java:150-162"),
    ("true,", None, MSG_TRAILING, "BooleanParsingTest.java:30-43"),
    ("false,", None, MSG_TRAILING, "BooleanParsingTest.java:30-43"),
    ("null,", None, MSG_TRAILING, "NullParsingTest.java:43-55"),
] + [
    (s, None, "Invalid value starting at %d. Expected 'false'." % s.index("f"), "BooleanParsingTest.java:45-58")
    for s in ["fals", "falsee", "[f]", '{"a":f}']
] + [
    (s, None, "Invalid value starting at %d. Expected 'true'." % s.index("t"), "BooleanParsingTest.java:60-73")
    for s in ["tru", "truee", "[t]", '{"a":t}']
] + [
    (s, None, "Invalid value starting at %d. Expected 'null'." % s.index("n"), "NullParsingTest.java:28-41")
    for s in ["[n]", '{"a":n}']
] + [
    (s, None, "Invalid value starting at 0. Expected 'null'.", "NullParsingTest.java:57-70") for s in ["nulll", "nul"]
] + [
    ("true", 3, "Invalid value starting at 0. Expected 'true'.", "BooleanParsingTest.java:93-105"),
    ("false", 4, "Invalid value starting at 0. Expected 'false'.", "BooleanParsingTest.java:107-119"),
    ("null", 3, "Invalid value starting at 0. Expected 'null'.", "NullParsingTest.java:92-104"),
]

# valid documents with the value the reference asserts through JsonValue accessors
VALID_DOCS = [
    ("[]", ("a", 0, []), "ArrayParsingTest.java:19-35"),
    ("[1, 2, 3]", ("a", 3, [("l", 1), ("l", 2), ("l", 3)]), "ArrayParsingTest.java:37-72"),
    ("{}", ("o", 0, []), "ObjectParsingTest.java:15-28"),
    ('{"a": 1, "b": 2, "c": 3}', ("o", 3, [(b"a", ("l", 1)), (b"b", ("l", 2)), (b"c", ("l", 3))]),
     "ObjectParsingTest.java:30-52"),
    ('{"ąćśńźż": 1, "\\u20A9\\u0E3F": 2, "αβγ": 3, "😀abc😀": 4}',
     ("o", 4, [("ąćśńźż".encode(), ("l", 1)), ("₩฿".encode(), ("l", 2)), ("αβγ".encode(), ("l", 3)),
               ("😀abc😀".encode(), ("l", 4))]), "ObjectParsingTest.java:68-82"),
    ('[{"a": 1}, {"a": 2}, {"a": 3}]',
     ("a", 3, [("o", 1, [(b"a", ("l", 1))]), ("o", 1, [(b"a", ("l", 2))]), ("o", 1, [(b"a", ("l", 3))])]),
     "ObjectParsingTest.java:113-133"),
    ("[true, false]", ("a", 2, [("t",), ("f",)]), "BooleanParsingTest.java:75-91"),
    ("[null, null, null]", ("a", 3, [("n",), ("n",), ("n",)]), "NullParsingTest.java:72-90"),
    ("true", ("t",), "BooleanParsingTest.java:16-28"),
    ("false", ("f",), "BooleanParsingTest.java:16-28"),
    ("null", ("n",), "NullParsingTest.java:15-26"),
]
