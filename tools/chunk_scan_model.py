"""Python model of the entry-state scan behind the chunk-parallel cooperative walker (csrc/coop_walk.hip: k_chunk_summary,
scan_apply, k_group_summary / k_top_scan / k_group_replay) -- the algebra only, no GPU: what a chunk of structurals does to
the walker's state is summarised relative to the chunk's start (net depth change, minimum depth, tape words, the containers
it opens and leaves open by relative level, the commas it adds to the innermost container it leaves untouched), applying a
summary to a state is associative, and the state in front of every chunk obtained that way equals the sequential walker's.
tests/test_chunk_scan_model.py checks exactly that on random bracket sequences; the kernels are then checked against the
oracle on the GPU (tests/test_gpu_coop_walk.py).

Tokens: '[' '{' ']' '}' ',' ':' 's' (string) 'n' (number: two tape words) 'a' (atom)."""
BIAS = 32
LEVELS = 64


def words_of(t):
    return 0 if t in ",:" else (2 if t == "n" else 1)


class State:
    """the walker's state in front of a structural: depth, tape position, per open level (tape position of the opening word,
    commas seen so far, is-array)"""

    def __init__(self):
        self.H = 0
        self.T = 1
        self.levels = {}  # level -> [tpos, commas, is_array]
        self.root_closed = False

    def copy(self):
        s = State()
        s.H, s.T, s.root_closed = self.H, self.T, self.root_closed
        s.levels = {k: list(v) for k, v in self.levels.items()}
        return s

    def key(self):
        return (self.H, self.T, self.root_closed, tuple(sorted((k, tuple(v)) for k, v in self.levels.items() if k < self.H)))


def step(state, t):
    """the sequential walker's bookkeeping for one structural (JsonIterator / TapeBuilder state, csrc/coop_walk.hip (2)-(4))"""
    if t in "[{":
        state.levels[state.H] = [state.T, 0, t == "["]
        state.H += 1
    elif t in "]}":
        if state.H > 0:
            state.H -= 1
            state.levels.pop(state.H, None)
        if state.H == 0:
            state.root_closed = True
    elif t == ",":
        if state.H > 0 and state.H - 1 in state.levels:
            state.levels[state.H - 1][1] += 1
    elif state.H == 0:
        state.root_closed = True  # a primitive root
    state.T += words_of(t)


class Summary:
    """k_chunk_summary: everything relative to the chunk's start (levels biased by BIAS)"""

    def __init__(self, tokens=()):
        self.delta = 0
        self.min_after = 10 ** 9
        self.words = 0
        self.exp = {}  # biased relative level -> [relative tpos, commas, is_array]
        self.low_commas = {}  # biased relative level -> commas added to a container from before the chunk
        self.out_of_range = False
        H = BIAS
        for t in tokens:
            if t in "[{":
                self.exp[H] = [self.words, 0, t == "["]
                H += 1
            elif t in "]}":
                H -= 1
                self.exp.pop(H, None)
                self.low_commas.pop(H, None)  # the container of this level (from before) closed: its commas are final
            elif t == ",":
                lvl = H - 1
                if lvl in self.exp:
                    self.exp[lvl][1] += 1
                else:
                    self.low_commas[lvl] = self.low_commas.get(lvl, 0) + 1
            self.words += words_of(t)
            self.min_after = min(self.min_after, H)
            if H - 1 < 0 or H >= LEVELS:
                self.out_of_range = True
        self.delta = H - BIAS
        if self.min_after == 10 ** 9:
            self.min_after = BIAS
        self.min_after -= BIAS
        m = min(0, self.min_after)
        # only the innermost untouched container's commas matter: level BIAS + m - 1
        low = BIAS + m - 1
        self.commas_low = self.low_commas.get(low, 0)
        self.exp = {k: v for k, v in self.exp.items() if BIAS + m <= k < BIAS + self.delta}


def apply(state, s):
    """scan_apply (absolute levels): the state behind the chunk from the state in front of it"""
    m = min(0, s.min_after)
    if state.H + s.min_after <= 0:
        state.root_closed = True
    low = state.H + m - 1
    if low in state.levels:
        state.levels[low][1] += s.commas_low
    for lvl in list(state.levels):
        if lvl >= state.H + m:
            del state.levels[lvl]
    for k, (tp, cnt, arr) in s.exp.items():
        state.levels[k - BIAS + state.H] = [state.T + tp, cnt, arr]
    state.H = max(0, state.H + s.delta)
    state.T += s.words


def entry_states(tokens, chunk, group):
    """the two-level scan: -> list of States in front of every chunk"""
    chunks = [tokens[i:i + chunk] for i in range(0, len(tokens), chunk)]
    sums = [Summary(c) for c in chunks]
    out = []
    s = State()
    del group  # (the kernels compose groups of summaries first -- applying is associative -- and replay them; same states)
    for k in range(len(chunks)):
        out.append(s.copy())
        apply(s, sums[k])
    return out, s
