"""Multi-GPU batched mode: shard documents across ranks, gather per-shard counts.

The path shards by DOCUMENT (independent units): rank r gets a contiguous, byte-balanced range of
documents, runs the same single-GPU kernels on it, and keeps its outputs local.  The only collective is one
all_gather of 4 x int64 per rank -- {documents, structurals, string bytes, failed documents} -- so that every rank
knows the global output offsets (north_star: "RCCL over xGMI only as a gather of per-shard counts").  On GPUs the
process group is "nccl" (= RCCL); the CPU tests run the same code over "gloo".
"""
import numpy as np


def partition_documents(doc_offsets, world_size):
    """Contiguous, byte-balanced document ranges: -> list of (first_doc, last_doc_exclusive) per rank.
    doc_offsets has n+1 entries (doc k = [offsets[k], offsets[k+1]))."""
    offs = np.asarray(doc_offsets, dtype=np.uint64)
    n = offs.size - 1
    total = int(offs[-1] - offs[0])
    bounds = [0]
    for r in range(1, world_size):
        target = int(offs[0]) + (total * r) // world_size
        k = int(np.searchsorted(offs, np.uint64(target), side="left"))
        k = min(max(k, bounds[-1]), n)
        bounds.append(k)
    bounds.append(n)
    return [(bounds[r], bounds[r + 1]) for r in range(world_size)]


def gather_counts(local_counts, device=None):
    """all_gather of the per-shard {docs, structurals, string_bytes, failed_docs}; -> int64 tensor [world, 4].
    With a single process (no initialised process group) returns the local row."""
    import torch
    import torch.distributed as dist
    row = torch.tensor([int(x) for x in local_counts], dtype=torch.int64, device=device)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return row[None, :]
    out = torch.empty(dist.get_world_size() * row.numel(), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(out, row)
    return out.view(dist.get_world_size(), row.numel())


def global_offsets(gathered):
    """Exclusive prefix over ranks of the gathered counts: row r = where rank r's outputs start globally."""
    import torch
    g = gathered.to(torch.int64)
    return torch.cumsum(g, dim=0) - g


class BatchShard:
    """One rank's contiguous share of a batch of documents, resident on its device, and everything one step of the
    batched path needs: outputs sized once, then `step()` queues sjmi_parse_batch_device (isolated stage 1 -> string
    records -> GPU walk; no host round trip) on the given stream.  `engine` is a binding.Context (or, in the CPU tests,
    a stub with the same parse_batch_device signature); tensors are plain torch tensors on `device`."""

    def __init__(self, engine, shard_bytes, local_offsets, device, max_depth=1024, index_ratio=1, string_ratio=1.0,
                 tape_ratio=1.0):
        import torch
        self.engine = engine
        self.device = device
        self.n = int(shard_bytes.numel()) if hasattr(shard_bytes, "numel") else len(shard_bytes)
        offs = np.ascontiguousarray(local_offsets, dtype=np.uint64)
        assert offs[0] == 0 and int(offs[-1]) == self.n  # (the documents cover the shard exactly: include/sjmi.h, sjmi_parse_batch_device)
        self.n_docs = offs.size - 1
        if hasattr(shard_bytes, "numel"):
            self.buf = torch.zeros(self.n + 128, dtype=torch.uint8, device=device)
            self.buf[:self.n] = shard_bytes
        else:
            self.buf = torch.zeros(self.n + 128, dtype=torch.uint8, device=device)
            self.buf[:self.n] = torch.frombuffer(bytearray(shard_bytes), dtype=torch.uint8).to(device)
        self.offs = torch.from_numpy(offs.view(np.int64).copy()).to(device)
        # the defaults cover the worst cases (one structural per byte: "[[[[", one tape word per byte: "[1,1,1"); real
        # JSON has one structural per 5-11 bytes, so a caller that knows its data passes tighter ratios (bench.py) -- a
        # shortfall is reported by the kernels (never overrun) and check() raises
        self.index_capacity = self.n // index_ratio + self.n_docs + 16
        self.idx = torch.empty(self.index_capacity, dtype=torch.int32, device=device)
        self.index_offsets = torch.zeros(self.n_docs + 1, dtype=torch.int64, device=device)
        self.doc_status = torch.zeros(max(self.n_docs, 1), dtype=torch.int32, device=device)
        self.sb_capacity = int(self.n * string_ratio) + 4 * self.index_capacity + 64
        self.sb = torch.empty(self.sb_capacity, dtype=torch.uint8, device=device)
        self.doc_string_offsets = torch.zeros(self.n_docs + 1, dtype=torch.int64, device=device)
        self.tape_capacity = int(self.n * tape_ratio) + 2 * self.n_docs + 8
        self.tape = torch.empty(self.tape_capacity, dtype=torch.int64, device=device)
        self.tape_offsets = torch.zeros(self.n_docs + 1, dtype=torch.int64, device=device)
        self.doc_errors = torch.zeros(max(self.n_docs, 1), dtype=torch.int32, device=device)
        self.result = torch.zeros(9, dtype=torch.int64, device=device)  # sjmi_batch_result
        self.max_depth = max_depth
        self._ndocs = torch.tensor([self.n_docs], dtype=torch.int64, device=device)

    def step(self, stream=0, exact=False, rejected=False):
        """One step, queued on `stream`.  Default: sjmi_parse_batch_device_optimistic -- only the optimistic pipeline (eight queue
        entries); a batch it cannot take (a document fails stage 1, a separator is missing) comes back with SJMI_ST_REJECTED in
        its result record and check() makes the call for rejected batches then (sjmi_parse_batch_device_rejected: per-document
        verdicts + the pipeline over a sanitized copy), off the hot path.  exact=True: sjmi_parse_batch_device, everything
        queued, every document decided on its own whatever the batch contains."""
        # Latched by check(): data that was rejected once does not take the optimistic-only call again (a step whose record says
        # REJECTED has no valid counts for the gather).  Rejected although every document passed stage 1 = rejected for its FORMAT
        # (no control-character separators): the next batch will be too, so it goes straight to the call for rejected batches;
        # rejected for a document that fails stage 1: the exact call, which tries the plain pass first.
        rejected = rejected or (not exact and getattr(self, "format_rejected", False) and not getattr(self, "unrepairable", False))
        exact = exact or getattr(self, "rejected_steps", 0) > 0
        if rejected and hasattr(self.engine, "parse_batch_device_rejected"):
            fn = self.engine.parse_batch_device_rejected
        else:
            fn = self.engine.parse_batch_device if exact or rejected or not hasattr(self.engine, "parse_batch_device_optimistic") \
                else self.engine.parse_batch_device_optimistic
        self._last_stream = stream
        fn(self.buf.data_ptr(), self.n, self.offs.data_ptr(), self.n_docs, self.idx.data_ptr(),
                                       self.index_capacity, self.index_offsets.data_ptr(), self.doc_status.data_ptr(),
                                       self.sb.data_ptr(), self.sb_capacity, self.doc_string_offsets.data_ptr(), self.max_depth,
                                       self.tape.data_ptr(), self.tape_capacity, self.tape_offsets.data_ptr(),
                                       self.doc_errors.data_ptr(), self.result.data_ptr(), stream)

    def counts_tensor(self):
        """The per-shard row of the count gather, on the device, without a host copy:
        {documents, structurals, string bytes, failed + handed-back documents}."""
        import torch
        r = self.result
        return torch.cat([self._ndocs, r[0:1], r[2:3], r[6:7] + r[7:8]])

    def check(self):
        """Host-side verdict of the last step (synchronise first): raises if a capacity was exceeded."""
        r = self.result.cpu().numpy()
        st1, sflags, wflags = int(r[1]) & 0xFFFFFFFF, int(r[4]) & 0xFFFFFFFF, int(r[8]) & 0xFFFFFFFF
        if st1 & 0x800:  # SJMI_ST_REJECTED: not a batch for the optimistic pipeline -- the exact call, here, off the hot path
            import torch
            self.rejected_steps = getattr(self, "rejected_steps", 0) + 1
            if not (st1 & 0xFF):
                self.format_rejected = True  # (a clean stage-1 verdict and still rejected: the separators)
            for entry in ({"rejected": True}, {"exact": True}):
                # the call for rejected batches (repair on the device); a batch it cannot take either -- documents that are not
                # separated AND let a scalar run on across a boundary -- says REJECTED once more and is the exact call's
                self.step(getattr(self, "_last_stream", 0), **entry)
                if str(self.device) != "cpu":  # (the CPU tests' stub engines are synchronous)
                    torch.cuda.synchronize(self.device)
                r = self.result.cpu().numpy()
                st1, sflags, wflags = int(r[1]) & 0xFFFFFFFF, int(r[4]) & 0xFFFFFFFF, int(r[8]) & 0xFFFFFFFF
                if not (st1 & 0x800):
                    break
                self.format_rejected = False  # (... and so are the batches behind it: the latch below moves to the exact call)
                self.unrepairable = True
        if st1 & 0x300:
            raise RuntimeError("stage 1 of the shard: capacity / internal error (status 0x%x)" % st1)
        if sflags & 1:
            raise RuntimeError("string buffer capacity exceeded")
        if sflags & 2:  # (the walkers read record offsets by string ordinal: with the table short they walk nothing)
            raise RuntimeError("more strings than the record table holds: index capacity exceeded")
        if sflags & 0xC:
            raise RuntimeError("string pass: engine fault (flags 0x%x)" % sflags)
        if wflags & 1:
            raise RuntimeError("tape capacity exceeded")
        return {"documents": self.n_docs, "structurals": int(r[0]), "string_bytes": int(r[2]), "tape_words": int(r[5]),
                "host_documents": int(r[6]), "failed_documents": int(r[7]), "stage1_status": st1 & 0xFF}


def sharded_step(shard, stream=0, always_gather=False):
    """One step of the multi-GPU batched path on this rank: the shard's kernels, then the count gather (the ONLY
    collective, north_star).  -> gathered [world, 4] int64 tensor on the shard's device.  always_gather: issue the
    collective even in a group of one rank (bench.py --sharded: the RCCL call path on a single GPU)."""
    import torch
    import torch.distributed as dist
    side = None
    if not stream and str(shard.device) != "cpu":
        # stream handle 0 means "the engine context's own stream" to the C ABI -- and it is also the handle of torch's legacy
        # default stream, so it cannot name that one.  The torch ops that read the result record below must be ordered behind
        # the kernels: the kernels go on a side stream of the shard's, fenced against the caller's current stream on both
        # sides with events (wait_stream) -- stream order only, no host synchronisation anywhere in the step.
        cur = torch.cuda.current_stream(shard.device)
        if cur.cuda_stream:
            stream = cur.cuda_stream
        else:
            side = getattr(shard, "_side_stream", None)
            if side is None:
                side = shard._side_stream = torch.cuda.Stream(device=shard.device)
            side.wait_stream(cur)
            stream = side.cuda_stream
    shard.step(stream)
    if side is not None:
        torch.cuda.current_stream(shard.device).wait_stream(side)
    row = shard.counts_tensor()
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not always_gather):
        return row[None, :]
    out = torch.empty(dist.get_world_size() * 4, dtype=torch.int64, device=shard.device)
    dist.all_gather_into_tensor(out, row)
    return out.view(dist.get_world_size(), 4)


# ---------------------------------------------------------------------------------------------------------------------
# ONE document over several GPUs / as a stream of chunks
# ---------------------------------------------------------------------------------------------------------------------
def split_points(n_bytes, parts, align=64):
    """Contiguous byte ranges [(a, b)] covering [0, n_bytes): every boundary a multiple of `align` (stage 1 works in
    64-byte blocks; a shard that is not the last one has no tail block), about equal sizes."""
    bounds = [0]
    for r in range(1, parts):
        b = (n_bytes * r // parts) // align * align
        bounds.append(max(b, bounds[-1]))
    bounds.append(n_bytes)
    return [(bounds[r], bounds[r + 1]) for r in range(parts)]


class HaloTooShort(RuntimeError):
    """SJMI_ST_HALO: a backslash run fills the whole left halo of a shard / chunk -- repeat with a larger one"""


class DocumentShard:
    """One rank's shard [a, b) of a document, resident on its device together with `halo` bytes of the document in front
    of it (the carries stage 1 needs are re-derived from them; only the in-string parity has to come from the ranks in
    front).  `data` = the document's bytes [a - halo, b) (bytes or a uint8 tensor)."""

    def __init__(self, engine, data, halo, is_last, device, index_ratio=1, halo_from_start=False):
        import torch
        self.engine, self.device, self.halo, self.is_last = engine, device, int(halo), bool(is_last)
        self.halo_from_start = bool(halo_from_start)  # the halo begins at the document's first byte
        total = int(data.numel()) if hasattr(data, "numel") else len(data)
        self.length = total - self.halo
        assert self.halo % 64 == 0 and (self.is_last or self.length % 64 == 0)
        self.buf = torch.zeros(total + 128, dtype=torch.uint8, device=device)
        self.buf[:total] = data if hasattr(data, "numel") else torch.frombuffer(bytearray(data), dtype=torch.uint8).to(device)
        self.capacity = self.length // index_ratio + 66
        self.idx = torch.empty(self.capacity, dtype=torch.int32, device=device)
        self.result = torch.zeros(2, dtype=torch.int64, device=device)  # sjmi_stage1_result
        self.entry_parity = 0

    def run(self, entry_parity, stream=0):
        self.entry_parity = int(entry_parity)
        self.engine.stage1_shard_device(self.buf.data_ptr() + self.halo, self.length, self.halo, self.is_last, self.entry_parity,
                                        self.idx.data_ptr(), self.capacity, self.result.data_ptr(), stream,
                                        halo_from_start=self.halo_from_start)

    def outcome(self):
        """(synchronise first) -> (count, status bits without UNCLOSED, parity after the shard)"""
        r = self.result.cpu().numpy()
        st = int(r[1]) & 0xFFFFFFFF
        if st & 0x300:
            raise RuntimeError("stage 1 of the shard: capacity / internal error (status 0x%x)" % st)
        if st & 0x400:
            raise HaloTooShort("a backslash run fills the %d bytes of halo in front of the shard" % self.halo)
        return int(r[0]), st & 0xFD, (st >> 1) & 1


def resolve_split_document(shard, sync):
    """The protocol of a document split over the ranks of the default process group (or a single process):
      1. every rank scans its shard as if it started outside a string;
      2. all_gather of one word per rank: does the shard flip the in-string parity?  (the first collective)
      3. a rank whose true entry parity is 1 (XOR of the flips in front of it) scans again with entry parity 1;
      4. all_gather of {count, status, parity after}: global index offsets, the document's verdict.
    `sync` = a callable that synchronises the rank's stream.  -> dict(entry_parity, count, offset, total, status)
    where status has SJMI_ST_UNCLOSED set iff the LAST shard ends inside a string."""
    import torch
    import torch.distributed as dist
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    rank = dist.get_rank() if multi else 0
    world = dist.get_world_size() if multi else 1
    shard.run(0)
    sync()
    _, _, flip = shard.outcome()  # entered with parity 0: the parity after IS the flip
    flips = torch.tensor([flip], dtype=torch.int64, device=shard.device)
    if multi:
        allf = torch.empty(world, dtype=torch.int64, device=shard.device)
        dist.all_gather_into_tensor(allf, flips)
    else:
        allf = flips
    entry = int(allf[:rank].sum().item()) & 1
    if entry:
        shard.run(1)
        sync()
    count, st, after = shard.outcome()
    row = torch.tensor([count, st, after], dtype=torch.int64, device=shard.device)
    if multi:
        allr = torch.empty(world * 3, dtype=torch.int64, device=shard.device)
        dist.all_gather_into_tensor(allr, row)
        allr = allr.view(world, 3)
    else:
        allr = row[None, :]
    a = allr.cpu().numpy()
    status = 0
    for r in range(world):
        status |= int(a[r, 1])
    if int(a[world - 1, 2]):
        status |= 2  # SJMI_ST_UNCLOSED: the document ends inside a string
    return {"entry_parity": entry, "count": count, "offset": int(a[:rank, 0].sum()), "total": int(a[:, 0].sum()), "status": status}


def stream_document(engine, device, chunks, halo=64):
    """Document-stream mode on ONE GPU: the chunks of a document (bytes; every chunk but the last a multiple of 64 long)
    are scanned one after the other, the in-string parity carried from chunk to chunk, each chunk seeing `halo` bytes of
    its predecessor.  -> (list of (base offset, indexes np.uint32 relative to the chunk), status)."""
    import numpy as np
    import torch
    out, status, parity, base, tail = [], 0, 0, 0, b""
    keep = max(halo, 4096)  # bytes of the stream kept for the halo (a chunk is repeated with more of them on SJMI_ST_HALO)
    for k, ch in enumerate(chunks):
        last = k == len(chunks) - 1
        want = halo
        while True:
            h = min(want, len(tail)) // 64 * 64
            sh = DocumentShard(engine, tail[len(tail) - h:] + ch if h else ch, h, last, device, halo_from_start=(h == base))
            sh.run(parity)
            torch.cuda.synchronize()
            try:
                count, st, after = sh.outcome()
                break
            except HaloTooShort:
                if h >= len(tail) // 64 * 64:
                    raise  # (everything that is kept of the stream is backslashes)
                want *= 4
        parity = after
        status |= st
        out.append((base, sh.idx[:count].cpu().numpy().view(np.uint32).copy()))
        base += len(ch)
        tail = (tail + ch)[-keep:]
    if parity:
        status |= 2
    return out, status
