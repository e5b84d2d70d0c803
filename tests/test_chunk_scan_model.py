"""The algebra of the chunk-parallel walker's entry-state scan (tools/chunk_scan_model.py, restating csrc/coop_walk.hip's
k_chunk_summary + scan_apply) against the sequential bookkeeping: the state in front of every chunk, for random bracket
sequences at every chunk length, including containers that span many chunks, commas credited to containers opened long
before, depth swings, roots that close early and documents that close too often."""
import os
import random
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import chunk_scan_model as M  # noqa: E402


def _random_tokens(rng, n, p_open, p_close, max_depth):
    out, depth = [], 0
    for _ in range(n):
        r = rng.random()
        if r < p_open and depth < max_depth:
            out.append(rng.choice("[{"))
            depth += 1
        elif r < p_open + p_close and depth > 0:
            out.append(rng.choice("]}"))
            depth -= 1
        else:
            out.append(rng.choice(",,,:sna"))
    return out


def test_entry_states_equal_the_sequential_walker():
    rng = random.Random(2026)
    checked = 0
    for it in range(400):
        n = rng.randint(1, 900)
        toks = ["["] + _random_tokens(rng, n, rng.choice([0.05, 0.15, 0.3]), rng.choice([0.05, 0.15, 0.3]), rng.choice([3, 12, 30]))
        for chunk in (1, 2, 7, 64, 128):
            entries, final = M.entry_states(toks, chunk, group=rng.choice([1, 3, 8]))
            s = M.State()
            for i, t in enumerate(toks):
                if i % chunk == 0:
                    assert entries[i // chunk].key() == s.key(), (it, chunk, i)
                    checked += 1
                M.step(s, t)
            assert final.key() == s.key(), (it, chunk)
    assert checked > 100000


def test_early_root_close_and_stray_brackets():
    for toks in (list("[s,s]") + list(",s,s"), list("[[s],[s]]") + list("]]s"), list("a") + list(",s"), list("{s:[n,n],s:{s:a}}")):
        for chunk in (1, 2, 3):
            entries, final = M.entry_states(toks, chunk, group=2)
            s = M.State()
            for i, t in enumerate(toks):
                if i % chunk == 0:
                    e = entries[i // chunk]
                    assert (e.H, e.T, e.root_closed) == (s.H, s.T, s.root_closed), (toks, chunk, i)
                M.step(s, t)
