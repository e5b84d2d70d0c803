"""Python model of csrc/unescape.hip's unescape_packed: the escaped strings of one wave as ONE packed byte stream, 64 virtual
bytes per window, the escape structure as bit algebra on per-window masks with carries from window to window.  It mirrors the
kernel's formulation (segment-start mask S, backslash-run carry left whole across string boundaries, the \\uXXXX digit masks
cut at the boundaries, per-string first error by position, output positions from a running count) so that the CPU suite can
check the algebra against the plain per-string unescape of the oracle (tests/test_packed_unescape_model.py); the kernel itself
is checked on the GPU (tests/test_gpu_unescape.py).  Codes: include/sjmi.h SJMI_E_* 4..8."""
EVEN = 0x5555555555555555
M64 = (1 << 64) - 1
ESC = {ord('"'): 0x22, ord("\\"): 0x5C, ord("/"): 0x2F, ord("b"): 8, ord("f"): 12, ord("n"): 10, ord("r"): 13, ord("t"): 9}


def hex4(b, p):
    v = 0
    for c in b[p:p + 4]:
        if 0x30 <= c <= 0x39:
            d = c - 0x30
        elif 0x61 <= (c | 0x20) <= 0x66:
            d = (c | 0x20) - 0x61 + 10
        else:
            return -1
        v = v << 4 | d
    return v if len(b[p:p + 4]) == 4 else -1


def unescape_packed(buf, strings):
    """buf: the document (bytes, padded); strings: [(open, close)] of the escaped strings, ascending.
    -> [bytes or -code] per string"""
    lens = [c - o - 1 for o, c in strings]
    voff, V = [], 0
    for n in lens:
        voff.append(V)
        V += n
    out = [bytearray() for _ in strings]
    err = [None] * len(strings)  # (window, lane, code) of the first error
    carry, pU, pD1, pD2, pD3, prev_U, cur = 0, 0, 0, 0, 0, 0, -1
    j_of = []
    for j, n in enumerate(lens):
        j_of += [j] * n
    for w0 in range(0, V, 64):
        lanes = range(64)
        valid = [w0 + t < V for t in lanes]
        sidx = [j_of[w0 + t] if valid[t] else cur for t in lanes]
        S = sum(1 << t for t in lanes if valid[t] and voff[sidx[t]] == w0 + t)
        pos = [strings[sidx[t]][0] + 1 + (w0 + t - voff[sidx[t]]) if valid[t] else 0 for t in lanes]
        c = [buf[pos[t]] if valid[t] else 0 for t in lanes]
        B = sum(1 << t for t in lanes if valid[t] and c[t] == 0x5C)
        bs = B & ~carry & M64
        follows = ((bs << 1) | carry) & M64
        odd_starts = bs & ~EVEN & ~follows & M64
        seq_even = odd_starts + bs
        carry_out = 1 if seq_even > M64 else 0
        seq_even &= M64
        escaped = (EVEN ^ (seq_even << 1)) & follows & M64
        U = sum(1 << t for t in lanes if (escaped >> t) & 1 and c[t] == ord("u") and valid[t])
        D1 = ((U << 1) | pU) & ~S & M64
        D2 = ((D1 << 1) | pD1) & ~S & M64
        D3 = ((D2 << 1) | pD2) & ~S & M64
        D4 = ((D3 << 1) | pD3) & ~S & M64
        digits = D1 | D2 | D3 | D4
        for t in lanes:
            if not valid[t]:
                continue
            j = sidx[t]
            is_esc = (escaped >> t) & 1
            is_start = ((B & ~escaped) >> t) & 1
            if is_start or (digits >> t) & 1:
                continue
            piece, e = bytes([c[t]]), 0
            if is_esc:
                if c[t] == ord("u"):
                    cp = hex4(buf, pos[t] + 1)
                    if 0xD800 <= cp <= 0xDBFF:
                        if buf[pos[t] + 5:pos[t] + 7] != b"\\u":
                            e = 7
                        else:
                            low = hex4(buf, pos[t] + 7) - 0xDC00
                            if low >> 10 != 0:
                                e = 8
                            else:
                                cp = (((cp - 0xD800) << 10) | low) + 0x10000
                    elif 0xDC00 <= cp <= 0xDFFF:
                        rel = w0 + t - voff[j]
                        paired = False
                        if rel >= 6:
                            prev_is_u = (U >> (t - 6)) & 1 if t >= 6 else (prev_U >> (58 + t)) & 1
                            if prev_is_u:
                                hi = hex4(buf, pos[t] - 5)
                                paired = 0xD800 <= hi <= 0xDBFF
                        if paired:
                            cp = -2
                        else:
                            e = 6
                    if not e:
                        if cp == -2:
                            piece = b""
                        elif cp < 0:
                            e = 5
                        else:
                            piece = chr(cp).encode("utf-8", "surrogatepass") if cp < 0x110000 else b""
                else:
                    r = ESC.get(c[t]) if c[t] < 0x80 else None
                    if r is None:
                        e = 4
                    else:
                        piece = bytes([r])
            if e:
                if err[j] is None:
                    err[j] = (w0, t, e)
                continue
            out[j] += piece
        cur = sidx[63] if valid[63] else cur
        pU, pD1, pD2, pD3 = U >> 63, D1 >> 63, D2 >> 63, D3 >> 63
        prev_U, carry = U, carry_out
    return [-err[j][2] if err[j] is not None else bytes(out[j]) for j in range(len(strings))]
