"""TEST INFRASTRUCTURE (see sj_oracle.h): a plain-Python restatement of the reference's on-demand cursor,
org.simdjson.OnDemandJsonIterator (/root/reference/src/main/java/org/simdjson/OnDemandJsonIterator.java:7-675), with the
pieces of NumberParser (NumberParser.java:199-310), StringParser (StringParser.java:25-161) and BitIndexes
(BitIndexes.java:47-101) it calls -- method for method, message for message.  Only tests/ may import it: it is the checker
for csrc/host/ondemand.h (the product's C++ mirror, which adds the GPU skip table).  Pure-Python loops: small documents.
Pinned by tests/test_host_ondemand.py (tests/golden/ondemand_vectors.py: what the reference's own schema-based tests assert).
Every getter of the class is restated."""
import struct


class JsonParsingException(Exception):
    pass


EMPTY, NULL, NOT_EMPTY = 0, 1, 2  # IteratorResult :672-674
_STRUCT_OR_WS = frozenset(b" \n\r\t,:[]{}")  # CharacterUtils.java:6-50
_ESCAPE = {ord('"'): 0x22, ord("\\"): 0x5C, ord("/"): 0x2F, ord("b"): 8, ord("f"): 12, ord("n"): 10, ord("r"): 13, ord("t"): 9}  # :52-72


def _jchar(b):
    """Java's (char) cast of a byte: sign-extended, so a byte >= 0x80 becomes the UTF-16 unit 0xFF80..0xFFFF"""
    return chr(b) if b < 0x80 else chr(0xFF00 | b)


class OnDemandJsonIterator:
    def __init__(self, buffer, length, indexes):
        """buffer: the document (bytes beyond `length` read as 0, the reference's padding); indexes: stage 1's structurals"""
        self.buffer = bytes(buffer[:length]) + b"\0" * 64
        self.len = length
        self.indexes = [int(x) for x in indexes]
        self.write_idx = len(self.indexes)
        self.read_idx = 0
        if self.write_idx == self.read_idx:  # init :34-41
            raise JsonParsingException("No structural element found.")
        self.depth = 1

    # ---- BitIndexes cursor (:47-80; the sentinel 0 behind the last index, :82-96) ----
    def _at(self, i):
        return self.indexes[i] if i < self.write_idx else 0

    def _get_and_advance(self):
        v = self._at(self.read_idx)
        self.read_idx += 1
        return v

    def _advance_and_get(self):
        self.read_idx += 1
        return self._at(self.read_idx)

    def _peek(self):
        return self._at(self.read_idx)

    def _has_next(self):
        return self.write_idx > self.read_idx

    def peek_byte(self):
        return self.buffer[self._peek()] if self._has_next() else 256

    # ---- skipChild :43-81 ----
    def skip_child(self, parent_depth=None):
        if parent_depth is None:
            parent_depth = self.depth - 1
        if self.depth <= parent_depth:
            return
        idx = self._get_and_advance()
        ch = self.buffer[idx]
        if ch in b"[{:,":
            pass
        elif ch == 0x22 and self.buffer[self._peek()] == 0x3A:
            self.read_idx += 1
        else:
            self.depth -= 1
            if self.depth <= parent_depth:
                return
        while self._has_next():
            ch = self.buffer[self._get_and_advance()]
            delta = 1 if ch in b"[{" else (-1 if ch in b"]}" else 0)
            self.depth += delta
            if delta < 0 and self.depth <= parent_depth:
                return
        raise JsonParsingException("Not enough close braces.")

    # ---- atoms :111-201 ----
    def _is(self, idx, word):
        return self.buffer[idx:idx + len(word)] == word

    def _visit_atom(self, idx, word, root):
        n = len(word)
        if root:
            valid = idx + n <= self.len and self._is(idx, word) and (idx + n == self.len or self.buffer[idx + n] in _STRUCT_OR_WS)
        elif word == b"null":
            valid = self._is(idx, word)  # visitNullAtom :134-138: the byte behind it is not looked at
        else:
            valid = self._is(idx, word) and self.buffer[idx + n] in _STRUCT_OR_WS
        if not valid:
            raise JsonParsingException("Invalid value starting at %d. Expected '%s'." % (idx, word.decode()))

    def get_boolean(self, root=False, nullable=True):
        """getRootNonNullBoolean :83-93, getRootBoolean :95-109, getNonNullBoolean :147-156, getBoolean :158-171 -> bool or None"""
        idx = self._get_and_advance()
        ch = self.buffer[idx]
        if ch == ord("t"):
            self._visit_atom(idx, b"true", root)
            result = True
        elif ch == ord("f"):
            self._visit_atom(idx, b"false", root)
            result = False
        elif ch == ord("n") and nullable:
            self._visit_atom(idx, b"null", root)
            result = None
        else:
            raise JsonParsingException("Unrecognized boolean value. Expected: 'true', 'false' or 'null'." if nullable
                                       else "Unrecognized boolean value. Expected: 'true' or 'false'.")
        if root:
            self.assert_no_more_json_values()
        self.depth -= 1
        return result

    # ---- numbers: getRoot[NonNull]Long :321-342, get[NonNull]Long :344-358, doubles :383-428 ----
    def _byte(self, q):
        # a root number is copied and padded with spaces (padRootNumber :406-412); any other number's byte behind it is
        # only looked at while `currentIdx < len` (NumberParser.java:219,:299)
        return self.buffer[q] if q < self.len else 0x20

    def get_long(self, root=False, nullable=True, bits=64):
        """bits = 8 / 16 / 32: the Byte :204-241, Short :243-280, Int :282-319 getters"""
        self.depth -= 1
        idx = self._get_and_advance()
        if nullable and self.buffer[idx] == ord("n"):
            self._visit_atom(idx, b"null", True)  # (visitRootNullAtom in the non-root form too, :354)
            if root:
                self.assert_no_more_json_values()
            return None
        value = self._parse_long(idx, bits)
        if root:
            self.assert_no_more_json_values()
        return value

    def get_double(self, root=False, nullable=True):
        self.depth -= 1
        idx = self._get_and_advance()
        if nullable and self.buffer[idx] == ord("n"):
            self._visit_atom(idx, b"null", True)  # (:424)
            if root:
                self.assert_no_more_json_values()
            return None
        value = self._parse_double(idx)
        if root:
            self.assert_no_more_json_values()
        return value

    def get_float(self, root=False, nullable=True):
        """getRootNonNullFloat :360-367, getRootFloat :369-381, getNonNullFloat :430-434, getFloat :436-444 -> the binary32
        value as a Python float (exactly), or None"""
        self.depth -= 1
        idx = self._get_and_advance()
        if nullable and self.buffer[idx] == ord("n"):
            self._visit_atom(idx, b"null", True)  # (:440)
            if root:
                self.assert_no_more_json_values()
            return None
        value = self._parse_double(idx, as_float=True)
        if root:
            self.assert_no_more_json_values()
        return value

    def _parse_long(self, offset, bits=64):  # NumberParser.parseByte :76-100, parseShort :115-139, parseInt :154-178, parseLong :199-224
        negative = self._byte(offset) == ord("-")
        cur = offset + 1 if negative else offset
        start = cur
        digits = 0
        while 0x30 <= self._byte(cur) <= 0x39:
            digits = (10 * digits + self._byte(cur) - 0x30) & 0xFFFFFFFFFFFFFFFF
            cur += 1
        count = cur - start
        if count == 0:
            raise JsonParsingException("Invalid number. Minus has to be followed by a digit.")
        if self._byte(start) == 0x30 and count > 1:
            raise JsonParsingException("Invalid number. Leading zeroes are not allowed.")
        if self._byte(cur) not in _STRUCT_OR_WS:
            raise JsonParsingException("Number has to be followed by a structural character or whitespace.")
        if bits == 64:
            if count > 19 or (count == 19 and not (negative and digits == 1 << 63) and digits >= 1 << 63):  # isOutOfLongRange :313-328
                raise JsonParsingException("Number value is out of long range ([-9223372036854775808, 9223372036854775807]).")
        else:  # isOutOfByteRange :102-113, isOutOfShortRange :141-152, isOutOfIntRange :180-191
            max_digits, max_abs = {8: 3, 16: 5, 32: 10}[bits], 1 << (bits - 1)
            if count > max_digits or (count == max_digits and digits > (max_abs if negative else max_abs - 1)):
                raise JsonParsingException("Number value is out of %s range ([%d, %d])." % ({8: "byte", 16: "short", 32: "int"}[bits], -max_abs, max_abs - 1))
        return -digits if negative else digits

    def _parse_double(self, offset, as_float=False):  # NumberParser.parseDouble :268-310 / parseFloat :226-266 (the same grammar)
        negative = self._byte(offset) == ord("-")
        cur = offset + 1 if negative else offset
        start = cur
        while 0x30 <= self._byte(cur) <= 0x39:
            cur += 1
        count = cur - start
        if count == 0:
            raise JsonParsingException("Invalid number. Minus has to be followed by a digit.")
        if self._byte(start) == 0x30 and count > 1:
            raise JsonParsingException("Invalid number. Leading zeroes are not allowed.")
        floating = False
        if self._byte(cur) == ord("."):
            floating = True
            cur += 1
            after = cur
            while 0x30 <= self._byte(cur) <= 0x39:
                cur += 1
            if cur == after:
                raise JsonParsingException("Invalid number. Decimal point has to be followed by a digit.")
        if self._byte(cur) in b"eE":
            floating = True
            cur += 1
            if self._byte(cur) in b"+-":
                cur += 1
            es = cur
            while 0x30 <= self._byte(cur) <= 0x39:
                cur += 1
            if cur == es:  # ExponentParser.java:27-29
                raise JsonParsingException("Invalid number. Exponent indicator has to be followed by a digit.")
        if not floating:
            raise JsonParsingException("Invalid floating-point number. Fraction or exponent part is missing.")
        if self._byte(cur) not in _STRUCT_OR_WS:
            raise JsonParsingException("Number has to be followed by a structural character or whitespace.")
        text = bytes(self._byte(q) for q in range(offset, cur)).decode()
        if as_float:
            return float32_of(text)  # FloatParser.java:62-330: the correctly rounded binary32
        return float(text)  # correctly rounded, saturating to +-inf / +-0: DoubleParser.java:79-330

    # ---- strings: getRootString :446-459, getString :461-472, getFieldName :646-652 ----
    def get_string(self, root=False):
        self.depth -= 1
        idx = self._get_and_advance()
        ch = self.buffer[idx]
        if ch == 0x22:
            out = self._parse_string(idx)
        elif ch == ord("n"):
            self._visit_atom(idx, b"null", root)
            out = None
        else:
            raise JsonParsingException("Invalid value starting at %d. Expected either string or 'null'." % idx)
        if root:
            self.assert_no_more_json_values()
        return out

    def get_char(self, root=False, nullable=True):
        """getNonNullChar :474-481, getChar :483-494, getRootNonNullChar :496-505, getRootChar :507-520 -> UTF-16 unit or None"""
        self.depth -= 1
        idx = self._get_and_advance()
        ch = self.buffer[idx]
        if ch == 0x22:
            out = self._parse_char(idx)
        elif nullable and ch == ord("n"):
            self._visit_atom(idx, b"null", root)
            out = None
        elif nullable:
            raise JsonParsingException("Invalid value starting at %d. Expected either string or 'null'." % idx)
        else:
            raise JsonParsingException("Invalid value starting at %d. Expected string." % idx)
        if root:
            self.assert_no_more_json_values()
        return out

    def _parse_char(self, start):  # StringParser.parseChar :70-110
        b = self.buffer
        i = start + 1
        if b[i] == 0x5C:
            e = b[i + 1]
            if e == ord("u"):
                cp = self._hex4(b, i + 2)
                if 0xD800 <= cp <= 0xDFFF:
                    raise JsonParsingException("Invalid code point. Should be within the range U+0000–U+D777 or U+E000–U+FFFF.")
                if cp < 0:
                    raise JsonParsingException("Invalid unicode escape sequence.")
                ch = cp
                i += 6
            else:
                r = _ESCAPE.get(e) if e < 0x80 else None
                if r is None:
                    raise JsonParsingException("Escaped unexpected character: " + _jchar(e))
                ch = r
                i += 2
        elif b[i] < 0x80:
            ch = b[i]
            i += 1
        elif b[i] & 0xE0 == 0xC0:
            ch = (b[i] & 0x1F) << 6 | (b[i + 1] & 0x3F)
            i += 2
        elif b[i] & 0xF0 == 0xE0:
            ch = (b[i] & 0x0F) << 12 | (b[i + 1] & 0x3F) << 6 | (b[i + 2] & 0x3F)
            i += 3
        else:
            raise JsonParsingException("String cannot be deserialized to a char. Expected a single 16-bit code unit character.")
        if b[i] != 0x22:
            raise JsonParsingException("String cannot be deserialized to a char. Expected a single-character string.")
        return ch & 0xFFFF

    def get_field_name(self):
        idx = self._get_and_advance()
        if self.buffer[idx] != 0x22:
            raise self._unexpected(idx, '"')
        return self._parse_string(idx)

    @staticmethod
    def _hex4(b, p):  # CharacterUtils.hexToInt :241-247
        v = 0
        for c in b[p:p + 4]:
            if 0x30 <= c <= 0x39:
                d = c - 0x30
            elif ord("a") <= (c | 0x20) <= ord("f"):
                d = (c | 0x20) - ord("a") + 10
            else:
                return -1
            v = v << 4 | d
        return v

    def _parse_string(self, idx):  # StringParser.doParseString :29-68 -> bytes
        b = self.buffer
        src = idx + 1
        out = bytearray()
        while True:
            c = b[src]
            if c == 0x22:
                return bytes(out)
            if c != 0x5C:
                out.append(c)
                src += 1
                continue
            e = b[src + 1]
            if e == ord("u"):
                cp = self._hex4(b, src + 2)
                src += 6
                if 0xD800 <= cp <= 0xDBFF:  # parseLowSurrogate :112-124
                    if b[src] != 0x5C or b[src + 1] != ord("u"):
                        raise JsonParsingException("Low surrogate should start with '\\u'")
                    low = self._hex4(b, src + 2) - 0xDC00
                    if low >> 10 != 0:
                        raise JsonParsingException("Invalid code point. Low surrogate should be in the range U+DC00–U+DFFF.")
                    cp = (((cp - 0xD800) << 10) | low) + 0x10000
                    src += 6
                elif 0xDC00 <= cp <= 0xDFFF:
                    raise JsonParsingException("Invalid code point. The range U+DC00–U+DFFF is reserved for low surrogate.")
                if cp < 0:
                    raise JsonParsingException("Invalid unicode escape sequence.")
                if cp <= 0x7F:  # storeCodePointInStringBuffer :126-161
                    out.append(cp)
                elif cp <= 0x7FF:
                    out += bytes([(cp >> 6) + 192, (cp & 63) + 128])
                elif cp <= 0xFFFF:
                    out += bytes([(cp >> 12) + 224, ((cp >> 6) & 63) + 128, (cp & 63) + 128])
                else:
                    out += bytes([(cp >> 18) + 240, ((cp >> 12) & 63) + 128, ((cp >> 6) & 63) + 128, (cp & 63) + 128])
            else:
                r = _ESCAPE.get(e) if e < 0x80 else None  # CharacterUtils.escape :74-83
                if r is None:
                    raise JsonParsingException("Escaped unexpected character: " + _jchar(e))
                out.append(r)
                src += 2

    # ---- containers :522-652 ----
    def _start(self, root, open_, close, unclosed, is_array):
        idx = self._peek()
        if self.buffer[idx] == ord("n"):
            self._visit_atom(idx, b"null", root)
            self.read_idx += 1
            self.depth -= 1
            return NULL
        if self.buffer[idx] != ord(open_):
            raise self._unexpected(idx, open_)
        if root and self.buffer[self.indexes[self.write_idx - 1]] != ord(close):
            raise JsonParsingException(unclosed)
        idx = self._advance_and_get()
        if self.buffer[idx] == ord(close):
            self.read_idx += 1
            self.depth -= 1
            if root:
                self.assert_no_more_json_values()
            return EMPTY
        if is_array:
            self.depth += 1
        return NOT_EMPTY

    def start_iterating_array(self, root=False):
        return self._start(root, "[", "]", "Unclosed array. Missing ']' for starting '['.", True)

    def start_iterating_object(self, root=False):
        return self._start(root, "{", "}", "Unclosed object. Missing '}' for starting '{'.", False)

    def next_array_element(self):
        ch = self.buffer[self._get_and_advance()]
        if ch == ord("]"):
            self.depth -= 1
            return False
        if ch == ord(","):
            self.depth += 1
            return True
        raise JsonParsingException("Missing comma between array values")

    def next_object_field(self):
        idx = self._get_and_advance()
        ch = self.buffer[idx]
        if ch == ord("}"):
            self.depth -= 1
            return False
        if ch == ord(","):
            return True
        raise self._unexpected(idx, ",")

    def move_to_field_value(self):
        idx = self._get_and_advance()
        if self.buffer[idx] != ord(":"):
            raise self._unexpected(idx, ":")
        self.depth += 1

    def _unexpected(self, idx, expected):  # :658-664
        if self.read_idx > self.write_idx:
            return JsonParsingException("Expected '%s' but reached end of buffer." % expected)
        return JsonParsingException("Expected '%s' but got: '%s'." % (expected, _jchar(self.buffer[idx])))

    def assert_no_more_json_values(self):  # :666-670
        if self._has_next():
            raise JsonParsingException("More than one JSON value at the root of the document, or extra characters at the end of the JSON!")


def float32_of(text):
    """the binary32 nearest to the decimal literal (ties to even; +-inf beyond the largest finite + half an ulp), computed
    exactly: candidates around the double-rounded value, compared with rational arithmetic"""
    from fractions import Fraction
    neg = text.startswith("-")
    body = text.lstrip("-").lower()
    mant, _, e = body.partition("e")
    exp = max(-100000, min(100000, int(e) if e else 0))
    ip, _, fp = mant.partition(".")
    digits = (ip + fp).lstrip("0")
    if not digits:
        return -0.0 if neg else 0.0
    lead = len(ip + fp) - len(digits)
    e10 = exp - len(fp)
    if len(digits) + e10 > 60:
        return float("-inf") if neg else float("inf")
    if len(digits) + e10 < -60:
        return -0.0 if neg else 0.0
    x = Fraction(int(digits)) * (Fraction(10) ** e10)
    del lead

    def f32(bits):
        return struct.unpack("<f", struct.pack("<I", bits))[0]
    lo, hi = 0, 0x7F800000  # positive binary32 bit patterns are ordered like their values
    while hi - lo > 1:
        mid = (lo + hi) // 2
        if Fraction(f32(mid)) <= x:
            lo = mid
        else:
            hi = mid
    below, above = Fraction(f32(lo)), (Fraction(f32(hi)) if hi < 0x7F800000 else Fraction(2) ** 128)
    if x - below < above - x or (x - below == above - x and lo % 2 == 0):
        pick = lo
    else:
        pick = hi
    v = f32(pick)
    return -v if neg else v


def float_bits(v):
    return struct.unpack("<I", struct.pack("<f", v))[0]


def double_bits(v):
    return struct.unpack("<Q", struct.pack("<d", v))[0]
