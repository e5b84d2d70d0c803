"""BatchShard's choice of entry point (sharding.py): the optimistic call on the hot path; behind SJMI_ST_REJECTED the call for rejected
batches, made by check(); afterwards a rejection for the batch's FORMAT (a clean stage-1 verdict and still rejected: the separators)
goes straight to that call, a rejection for a document that fails stage 1 to the exact call (which tries the plain pass first).
CPU test with a recording stub in place of binding.Context -- the kernels need a GPU; this is the host logic around them."""
import ctypes as C

import numpy as np
import torch


class _Recorder:
    """Writes the result record a real engine would: stage-1 status per call from a script, records which entry was called."""

    def __init__(self, statuses):
        self.statuses = list(statuses)   # status word of the optimistic call, in call order
        self.calls = []

    def _write(self, d_result, status):
        res = (C.c_int64 * 9).from_address(d_result)
        for i in range(9):
            res[i] = 0
        res[1] = status

    def _entry(name):
        def f(self, d_buf, total_len, d_doc_offsets, n_docs, d_indexes, index_capacity, d_index_offsets, d_doc_status, d_sb,
              sb_capacity, d_doc_string_offsets, max_depth, d_tape, tape_capacity, d_tape_offsets, d_doc_errors, d_result, stream=0):
            self.calls.append(name)
            self._write(d_result, self.statuses.pop(0) if name == "optimistic" else 0)
        return f

    parse_batch_device_optimistic = _entry("optimistic")
    parse_batch_device = _entry("exact")
    parse_batch_device_rejected = _entry("rejected")


def _shard(engine):
    from simdjson_java_amd import sharding
    docs = [b'{"a":1}\n', b"[1,2]\n", b"3\n"]
    buf = b"".join(docs)
    offs = np.cumsum([0] + [len(d) for d in docs]).astype(np.uint64)
    return sharding.BatchShard(engine, buf, offs, torch.device("cpu"))


def _step_and_check(shard):
    shard.step()
    return shard.check()


def test_accepted_batches_stay_on_the_optimistic_call():
    eng = _Recorder([0, 0, 0])
    sh = _shard(eng)
    for _ in range(3):
        _step_and_check(sh)
    assert eng.calls == ["optimistic"] * 3 and getattr(sh, "rejected_steps", 0) == 0


def test_a_content_rejection_is_repaired_and_latches_the_exact_call():
    eng = _Recorder([0x800 | 1])  # REJECTED with a UTF-8 verdict: a document fails stage 1
    sh = _shard(eng)
    _step_and_check(sh)
    assert eng.calls == ["optimistic", "rejected"] and sh.rejected_steps == 1 and not getattr(sh, "format_rejected", False)
    _step_and_check(sh)  # (a record that says REJECTED has no counts for the gather: such data does not take the optimistic-only call again)
    assert eng.calls[2:] == ["exact"]


def test_a_format_rejection_goes_straight_to_the_call_for_rejected_batches():
    eng = _Recorder([0x800])      # REJECTED with a clean stage-1 verdict: the separators
    sh = _shard(eng)
    _step_and_check(sh)
    assert eng.calls == ["optimistic", "rejected"] and sh.format_rejected
    _step_and_check(sh)
    _step_and_check(sh)
    assert eng.calls[2:] == ["rejected", "rejected"]
    sh.step(exact=True)           # (an explicit exact step is still the exact call)
    assert eng.calls[-1] == "exact"


def test_an_engine_without_the_new_entry_falls_back_to_the_exact_call():
    class Old:  # (round 5's binding: two entries)
        def __init__(self):
            self.inner = _Recorder([0x800])
            self.calls = self.inner.calls
            self.parse_batch_device_optimistic = self.inner.parse_batch_device_optimistic
            self.parse_batch_device = self.inner.parse_batch_device
    eng = Old()
    sh = _shard(eng)
    _step_and_check(sh)
    assert eng.calls == ["optimistic", "exact"]
